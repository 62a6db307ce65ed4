"""Shared blind rotations on a one-gate-one-bootstrap program: an n-bit ripple-carry adder built gate by
gate, lowered like the reference's MapToFBSBasic (netlist.map_basic: every two-input gate = one linear combination + one
table, so XOR and AND of the same pair of wires are two tables on one source), evaluated on T samples with and without
FBS_LOAD_FUSE_TABLES at the 128-bit parameter set chosen for each.  Every output is decrypted and checked.

    python3 tools/fusion_bench.py [bits=64] [T=1000]
"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from tfhe_fbs_map_amd import ExecConfig, security_bits                       # noqa: E402
from tfhe_fbs_map_amd.fbs_exec_env import min_fbs_size                      # noqa: E402
from tfhe_fbs_map_amd.netlist import BitExecEnv, map_basic                  # noqa: E402
from tfhe_fbs_map_amd.params import margin_sigmas                           # noqa: E402


def ripple_adder(bits):
    env = BitExecEnv()
    a = [env.input("a%d" % i) for i in range(bits)]
    b = [env.input("b%d" % i) for i in range(bits)]
    carry = None
    for i in range(bits):
        x = env.op_xor(a[i], b[i])
        g = env.op_and(a[i], b[i])
        if carry is None:
            s, carry = x, g
        else:
            s = env.op_xor(x, carry)
            carry = env.op_or(g, env.op_and(x, carry))
        env.output("s%d" % i, s)
    env.output("cout", carry)
    return env


def main():
    bits = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    lut = map_basic(ripple_adder(bits))
    low = lut.lower()
    p = min_fbs_size(low["tables"])
    rng = np.random.default_rng(3)
    x, y = rng.integers(0, 2, (bits, T)), rng.integers(0, 2, (bits, T))
    ins = {("a%d" % i): x[i] for i in range(bits)} | {("b%d" % i): y[i] for i in range(bits)}
    want = sum((x[i].astype(object) + y[i].astype(object)) << i for i in range(bits))
    out = {"adder_bits": bits, "samples": T, "p": p, "stats": lut.stats(), "fusion_stats": lut.fusion_stats(p)}
    for fuse in (False, True):
        cfg = ExecConfig(seed=9, fuse_tables=fuse)
        ctx, fused = cfg.choose(lut, p)
        prog = cfg.program_for(ctx, low, fused)
        cts = ctx.encrypt(np.stack([ins[n] for n in low["input_names"]]).astype(np.int64), nonce0=1)
        prog.eval(cts[:, :8].copy(), 8)                                      # warm-up: buffers, kernels
        t0 = time.perf_counter()
        res = prog.eval(cts, T)
        dt = time.perf_counter() - t0
        dec = ctx.decrypt(res)
        got = sum(dec[low["out_names"].index("s%d" % i)].astype(object) << i for i in range(bits))
        got = got + (dec[low["out_names"].index("cout")].astype(object) << bits)
        assert all(g == w for g, w in zip(got, want)), "wrong sum"
        prm = ctx.params
        norm2 = out["fusion_stats" if fused else "stats"]["norm2_linprod"]
        out["fused" if fuse else "plain"] = {
            "fused": fused, "seconds": round(dt, 3), "rotations": prog.n_rotations, "bootstraps": prog.n_bootstrap,
            "gate_bootstraps_per_s": round(prog.n_bootstrap * T / dt), "rotations_per_s": round(prog.n_rotations * T / dt),
            "params": dict(n=prm.n, N=prm.N, k=prm.k, l=prm.l_bsk, beta=prm.beta_bsk, t=prm.t_ksk, gamma=prm.gamma_ksk, bsk_group=prm.bsk_group),
            "security_bits": round(security_bits(prm), 1), "margin_sigmas": round(margin_sigmas(prm, norm2), 2), "all_sums_correct": True}
        ctx.close()
    out["speedup"] = round(out["plain"]["seconds"] / out["fused"]["seconds"], 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
