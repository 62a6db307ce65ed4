"""Which (k, N, l, key bits per step) the general-GLWE kernel k_blind_rotate_glwe makes available beside the shipped choices, and what
they cost: for a few (p, norm2) the cheapest 128-bit set per (k, N) by a plain instruction model, timed on the GPU at full batches
next to the set `choose_params` returns today.     python3 tools/glwe_candidates.py [batch = 1024] [steps = 4]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tfhe_fbs_map_amd import Params, choose_params
from tfhe_fbs_map_amd.params import _GADGETS, _SWITCHES, _switch_fits, margin_sigmas, security_bits
from tfhe_fbs_map_amd.security import MODULUS, sigma_min


def instr(n, N, log_n, k, l, group):
    """FP64 instructions per bootstrap / 64 lanes: transforms (8 per butterfly), products (7), bundle (3 x 7 per key word)"""
    steps = n / group
    per_coef = (k + 1) * ((l + 1) * 4.0 * log_n + 7.0 * (k + 1) * l * (4 if group == 2 else 1) + 2.0 * l + 12.0)
    return steps * N * per_coef / 64.0


def best_for(p, norm2, k, log_n, min_margin=6.0, groups=(1, 2)):
    q = float(MODULUS)
    N = 1 << log_n
    ns = np.arange(450, 1200, 2, dtype=np.float64)
    s_lwe = np.array([sigma_min(int(n), 128) for n in ns]) / q
    s_glwe = sigma_min(k * N, 128) / q
    need = (1.0 / (4.0 * p) / min_margin) ** 2
    v_ms = (1 + ns / 4.0) / (12.0 * (2.0 * N) ** 2)
    best = None
    for (l, beta) in _GADGETS:
        if (k + 1) * l > 20:
            continue
        for group in groups:
            B = 2.0 ** beta
            key_term = (k + 1) * l * N * (B * B + 2) / 12.0 * s_glwe ** 2
            round_term = (1 + k * N / 2.0) / (12.0 * B ** (2 * l))
            v_br = ns * (key_term + 0.5 * round_term) if group == 1 else ns / 2.0 * (6.0 * key_term + 1.5 * round_term)
            room = need - v_ms - norm2 * v_br
            for t, g in _SWITCHES:
                if not _switch_fits(t, g, k * N):
                    continue
                b2 = 2.0 ** g
                v_ks = k * N * (t * (b2 * b2 + 2) / 12.0 * s_lwe ** 2 + 0.5 / (12.0 * b2 ** (2 * t)))
                ok = np.nonzero(v_ks <= room)[0]
                if ok.size == 0:
                    continue
                n = int(ns[ok[0]])
                cost = instr(n, N, log_n, k, l, group) * (1 + 0.012 * t / 8.0)
                if best is None or cost < best[0]:
                    best = (cost, Params(n=n, log_n_poly=log_n, k=k, l_bsk=l, beta_bsk=beta, t_ksk=t, gamma_ksk=g, p_msg=p,
                                         sigma_lwe=int(round(s_lwe[ok[0]] * q)), sigma_glwe=int(round(s_glwe * q)), bsk_group=group))
    return best


def time_set(prm, B, steps):
    import torch
    from tfhe_fbs_map_amd import Context
    ctx = Context(prm, seed=1)
    rng = np.random.default_rng(42)
    p = prm.p_msg
    tables = [[0] + [int(v) for v in rng.integers(0, 2, p - 1)] for _ in range(16)]
    tv = ctx.tvset(tables)
    msgs = rng.integers(0, p, B)
    ids = (np.arange(B) % 16).astype(np.uint32)
    d_in = torch.from_numpy(ctx.encrypt(msgs, nonce0=0).view(np.int64)).cuda()
    d_ids = torch.from_numpy(ids.view(np.int32)).cuda()
    d_out = torch.empty_like(d_in)
    for _ in range(2):
        ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), B, d_out.data_ptr())
    ctx.sync()
    ctx.profile(True); ctx.profile_read(reset=True)
    for _ in range(steps):
        ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), B, d_out.data_ptr())
    ctx.sync()
    prof = ctx.profile_read()
    ok = bool(np.array_equal(ctx.decrypt(d_out.cpu().numpy().view(np.uint64)), [tables[i][m] for i, m in zip(ids, msgs)]))
    ms = (prof["blind_rotate"]["ms"] + prof["keyswitch"]["ms"]) / steps
    ctx.close()
    return ms, prof["blind_rotate"]["kernel"], ok


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    gpu = "--no-gpu" not in sys.argv
    for p, norm2 in ((2, 1), (3, 2), (4, 2), (7, 10), (15, 70)):
        ref = choose_params(p, norm2, glwe_dims=(1, 2))
        rows = [("shipped", instr(ref.n, ref.N, ref.log_n_poly, ref.k, ref.l_bsk, ref.bsk_group), ref)]
        for k in (1, 2, 3, 4):
            for log_n in (8, 9, 10):
                if k == 4 and log_n == 10:
                    continue
                b = best_for(p, norm2, k, log_n)
                if b:
                    rows.append(("k=%d N=%d" % (k, 1 << log_n), b[0], b[1]))
        rows.sort(key=lambda r: r[1])
        print("== p = %d, norm2 = %g" % (p, norm2), flush=True)
        for label, cost, prm in rows[:6] if rows[0][0] == "shipped" else rows[:5] + [r for r in rows if r[0] == "shipped"]:
            line = "  %-12s n=%d N=%d k=%d l=%d beta=%d bits/step=%d t=%d g=%d  model %.0f instr  margin %.1f sec %.0f" % (
                label, prm.n, prm.N, prm.k, prm.l_bsk, prm.beta_bsk, prm.bsk_group, prm.t_ksk, prm.gamma_ksk, cost, margin_sigmas(prm, norm2), security_bits(prm))
            if gpu and (prm.k >= 2 or label == "shipped") and not (prm.k == 1 and prm.bsk_group == 2 and prm.log_n_poly < 10):
                try:
                    ms, kern, ok = time_set(prm, B, steps)
                    line += "  | %.3f ms per %d = %.1f k FBS/s %s%s" % (ms, B, B / ms, kern, "" if ok else " WRONG")
                except Exception as exc:
                    line += "  | %s" % str(exc)[:80]
            print(line, flush=True)
