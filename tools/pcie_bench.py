"""PCIe-inclusive rate of the flat batch: host buffers through fbs_bootstrap_batch (H2D + key switch + blind rotation + D2H per
call, pageable numpy arrays) beside the device-resident call bench.py times.    python3 tools/pcie_bench.py [batch] [steps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                                          # noqa: E402
import torch                                                                # noqa: E402
from tfhe_fbs_map_amd import P1024, Context                                 # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
prm = P1024
if os.environ.get("SECURE"):                  # the default 128-bit set for p = 15 (GLWE dimension 2: ciphertexts of 2 N + 1 words)
    from tfhe_fbs_map_amd import choose_params
    prm = choose_params(15, 70, glwe_dims=(1, 2))
ctx = Context(prm, seed=1)
rng = np.random.default_rng(42)
tables = [[0] + [int(v) for v in rng.integers(0, 2, 14)] for _ in range(16)]
tv = ctx.tvset(tables)
msgs = rng.integers(0, 15, B)
ids = (np.arange(B) % 16).astype(np.uint32)
cts = ctx.encrypt(msgs, nonce0=0)
for _ in range(2):
    out = ctx.bootstrap_batch(tv, cts, ids)
t0 = time.perf_counter()
for _ in range(steps):
    out = ctx.bootstrap_batch(tv, cts, ids)
host = (time.perf_counter() - t0) / steps
d_in = torch.from_numpy(cts.view(np.int64)).cuda()
d_ids = torch.from_numpy(ids.view(np.int32)).cuda()
d_out = torch.empty_like(d_in)
for _ in range(2):
    ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), B, d_out.data_ptr())
ctx.sync()
t0 = time.perf_counter()
for _ in range(steps):
    ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), B, d_out.data_ptr())
ctx.sync()
dev = (time.perf_counter() - t0) / steps
ok = bool(np.array_equal(ctx.decrypt(out), [tables[i][m] for i, m in zip(ids, msgs)]))
print(json.dumps(dict(batch=B, steps=steps, host_buffers_fbs_per_s=round(B / host), device_buffers_fbs_per_s=round(B / dev),
                      host_ms_per_call=round(host * 1e3, 3), device_ms_per_call=round(dev * 1e3, 3),
                      bytes_moved_per_call=2 * cts.nbytes, decrypt_ok=ok)))
