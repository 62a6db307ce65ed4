"""The k = 2 latency shape of the library named by $FBS_LIB (kernel-variant experiments, tools/build_variants.sh): a spot check
against the oracle at toy n, then per-launch times at the shipped 128-bit set.   python3 tools/k2_latency.py [shape = 12] [steps = 6] [sizes ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tfhe_fbs_map_amd import Context, Params, choose_params
from oracle import tfhe_oracle as orc

shape = int(sys.argv[1]) if len(sys.argv) > 1 else 12
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
sizes = [int(v) for v in sys.argv[3:]] or [64, 256, 512]
toy = Params(n=16, log_n_poly=10, k=2, l_bsk=1, beta_bsk=21, t_ksk=8, gamma_ksk=2, p_msg=7, sigma_lwe=1 << 8, sigma_glwe=4, bsk_group=2)
ctx, o = Context(toy, seed=4), orc.Oracle(toy, seed=4)
tabs = [[0, 1, 1, 0, 1, 0, 0], [0, 1, 1, 0, 1, 0, 0, 1, 0, 0, 1, 0, 1, 1], [0, 1, 2, 3, 2, 1, 0]]
msgs = np.arange(23) % 7
ids = (np.arange(23) % 3).astype(np.uint32)
cts = ctx.encrypt(msgs, 3)
cts[22, :-1] = 0
ctx.tune(br_k2_shape=shape)
exact = bool(np.array_equal(ctx.bootstrap_batch(ctx.tvset(tabs), cts, ids), o.bootstrap_batch(cts, tabs, ids)[0]))
ctx.close()
prm = choose_params(15, 70, glwe_dims=(1, 2))
ctx = Context(prm, seed=1)
ctx.tune(br_k2_shape=shape)
rng = np.random.default_rng(42)
tables = [[0] + [int(v) for v in rng.integers(0, 2, 14)] for _ in range(16)]
tv = ctx.tvset(tables)
line = "%-44s exact=%s" % (os.environ.get("FBS_LIB", "in-tree").split("/")[-1], exact)
for B in sizes:
    msgs = rng.integers(0, 15, B)
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
    line += "  B=%d %.3f ms%s" % (B, prof["blind_rotate"]["ms"] / steps, "" if ok else " WRONG")
print(line + "  (%s)" % prof["blind_rotate"]["kernel"], flush=True)
