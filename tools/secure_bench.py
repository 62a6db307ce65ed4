"""A 128-bit-secure parameter set (params.choose_params(p, norm2); default p = 15, norm2 = 70: what the `secure` leg of bench.py
times) on a flat batch, alone, for rocprofv3 --pmc passes:  secure_bench.py [batch] [steps] [p] [norm2] [key bits per step | k2 | k3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tfhe_fbs_map_amd import Context, choose_params

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
p_msg = int(sys.argv[3]) if len(sys.argv) > 3 else 15
norm2 = float(sys.argv[4]) if len(sys.argv) > 4 else 70
if len(sys.argv) > 5 and sys.argv[5] == "k2":       # GLWE dimension 2 admitted (N = 1024, two key bits per step)
    prm = choose_params(p_msg, norm2, glwe_dims=(1, 2))
elif len(sys.argv) > 5 and sys.argv[5] == "k3":     # ... and 3 (N = 512): what ExecConfig() admits
    prm = choose_params(p_msg, norm2, glwe_dims=(1, 2, 3))
else:
    prm = choose_params(p_msg, norm2, groups=(int(sys.argv[5]),)) if len(sys.argv) > 5 else choose_params(p_msg, norm2)
ctx = Context(prm, seed=1)
rng = np.random.default_rng(42)
tables = [[0] + [int(v) for v in rng.integers(0, 2, p_msg - 1)] for _ in range(16)]
tv = ctx.tvset(tables)
msgs = rng.integers(0, p_msg, B)
ids = (np.arange(B) % 16).astype(np.uint32)
d_in = torch.from_numpy(ctx.encrypt(msgs, nonce0=0).view(np.int64)).cuda()
d_ids = torch.from_numpy(ids.view(np.int32)).cuda()
d_out = torch.empty_like(d_in)
for _ in range(2):
    ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), B, d_out.data_ptr())
ctx.sync()
ctx.profile(True); ctx.profile_read()
t0 = time.perf_counter()
for _ in range(steps):
    ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), B, d_out.data_ptr())
ctx.sync()
dt = time.perf_counter() - t0
prof = ctx.profile_read()
ok = np.array_equal(ctx.decrypt(d_out.cpu().numpy().view(np.uint64)), [tables[i][m] for i, m in zip(ids, msgs)])
print("n=%d N=%d k=%d l=%d beta=%d  %.0f FBS/s  br %.3f ms (%s)  ks %.3f ms  ok=%s" % (
    prm.n, prm.N, prm.k, prm.l_bsk, prm.beta_bsk, B * steps / dt, prof["blind_rotate"]["ms"] / steps, prof["blind_rotate"]["kernel"],
    prof["keyswitch"]["ms"] / steps, ok))
