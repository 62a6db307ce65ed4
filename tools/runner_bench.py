"""Overhead check: the Python level runners (device-pointer C ABI, per-level calls) against fbs_eval, one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.helpers import load_fixture, subsample
from tfhe_fbs_map_amd import P1024, Context, parse_fbs, _native as nat
from tfhe_fbs_map_amd.distributed import GateShardedRunner, GpuBackend, SampleShardedRunner

rec = load_fixture("mul16__search_p15"); T = 200
env = parse_fbs(rec["fbs"], inputs=rec["program_inputs"]); low = env.lower()
ins, expect = subsample(rec, T)
ctx = Context(P1024, seed=1); tv = ctx.tvset(low["tables"])
prog = nat.Program(ctx, tv, len(low["input_names"]), low["kind"], low["arg0"], low["arg1"], low["const_coef"],
                   low["term_coef"], low["term_src"], low["out_wire"])
cts = ctx.encrypt(np.stack([ins[n] for n in low["input_names"]]))
prog.eval(cts[:, :2].copy(), 2)
t0 = time.time(); ref = prog.eval(cts, T); t_prog = time.time() - t0
nf = prog.n_bootstrap * T
print("fbs_eval            %.2fs  %.0f FBS/s" % (t_prog, nf / t_prog))
for cls in (GateShardedRunner, SampleShardedRunner):
    r = cls(GpuBackend(prog)); r.run(cts[:, :2].copy(), 2)
    t0 = time.time(); out = r.run(cts, T); torch.cuda.synchronize(); dt = time.time() - t0
    print("%-20s %.2fs  %.0f FBS/s  identical=%s" % (cls.__name__, dt, nf / dt, np.array_equal(out, ref)))
# host-buffer flat batch (PCIe inclusive)
rng = np.random.default_rng(0); B = 1024
tabs = [[0] + [int(v) for v in rng.integers(0, 2, 14)] for _ in range(16)]
tv2 = ctx.tvset(tabs); c2 = ctx.encrypt(rng.integers(0, 15, B)); ids = (np.arange(B) % 16).astype(np.uint32)
ctx.bootstrap_batch(tv2, c2, ids)
t0 = time.time()
for _ in range(5): ctx.bootstrap_batch(tv2, c2, ids)
dt = (time.time() - t0) / 5
print("host-buffer batch of 1024 (PCIe both ways + alloc): %.2f ms -> %.0f FBS/s" % (dt * 1e3, B / dt))
