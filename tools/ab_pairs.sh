#!/bin/bash
# A/B of the two-bits-per-step kernel variants in gpurun_exp/ on the 128-bit set (timing; variant 2 computes garbage)
for lib in tfhe_fbs_map_amd/libfbsexec.so gpurun_exp/*.so; do
  echo "== $lib"
  FBS_LIB=$PWD/$lib timeout -k 10 120 python3 - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tfhe_fbs_map_amd import Context, choose_params
prm = choose_params(15, 70).replace(bsk_group=2)
ctx = Context(prm, seed=1)
B = 1024
rng = np.random.default_rng(42)
tabs = [[0] + [int(v) for v in rng.integers(0, 2, 14)] for _ in range(16)]
tv = ctx.tvset(tabs)
m = rng.integers(0, 15, B); idv = (np.arange(B) % 16).astype(np.uint32)
d_in = torch.from_numpy(ctx.encrypt(m, nonce0=0).view(np.int64)).cuda(); d_ids = torch.from_numpy(idv.view(np.int32)).cuda(); d_out = torch.empty_like(d_in)
for _ in range(2): ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), B, d_out.data_ptr())
ctx.sync(); ctx.profile(True); ctx.profile_read()
for _ in range(5): ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), B, d_out.data_ptr())
ctx.sync(); prof = ctx.profile_read()
ok = np.array_equal(ctx.decrypt(d_out.cpu().numpy().view(np.uint64)), [tabs[i][x] for i, x in zip(idv, m)])
print("br %.3f ms (%s) ok=%s" % (prof["blind_rotate"]["ms"] / 5, prof["blind_rotate"]["kernel"], ok))
PY
done
