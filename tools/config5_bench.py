"""BASELINE config 5: p = 31 on N = 2048 (params_for(31)), flat batch of 1024; timing + decrypt check."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tfhe_fbs_map_amd import Context, params_for
prm = params_for(31)
ctx = Context(prm, seed=1)
rng = np.random.default_rng(1)
tables = [[0] + [int(v) for v in rng.integers(0, 2, 30)] for _ in range(16)]
B = 1024
msgs = rng.integers(0, 31, B); ids = (np.arange(B) % 16).astype(np.uint32)
cts = ctx.encrypt(msgs)
tv = ctx.tvset(tables)
d_in = torch.from_numpy(cts.view(np.int64)).cuda(); d_ids = torch.from_numpy(ids.view(np.int32)).cuda(); d_out = torch.empty_like(d_in)
for _ in range(2): ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), B, d_out.data_ptr())
torch.cuda.synchronize(); ctx.profile(True); ctx.profile_read()
t0 = time.perf_counter()
for _ in range(5): ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), B, d_out.data_ptr())
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
p = ctx.profile_read()
ok = np.array_equal(ctx.decrypt(d_out.cpu().numpy().view(np.uint64)), [tables[i][m] for i, m in zip(ids, msgs)])
print("N=2048 l=%d beta=%d p=31: %.0f FBS/s  (br %.2f ms, ks %.2f ms per 1024)  decrypt_ok=%s" % (
    prm.l_bsk, prm.beta_bsk, B / dt, p["blind_rotate"]["ms"] / 5, p["keyswitch"]["ms"] / 5, ok))
