"""The k = 1 whole-CU kernels of the library named by $FBS_LIB (kernel-variant experiments, tools/build_variants.sh): a spot check
against the oracle at toy n per kernel family, then per-launch times at the real sets -- P1024 (k_blind_rotate_cu<10,3,2>), the
k = 1 128-bit set for p = 15 (k_blind_rotate_cu_pairs<11,1>) and the one for p = 31 (k_blind_rotate_cu_pairs<11,2>, every size).
    python3 tools/cu_latency.py [steps = 6]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tfhe_fbs_map_amd import Context, P1024, Params, choose_params
from oracle import tfhe_oracle as orc

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
tabs = [[0, 1, 1, 0, 1, 0, 0], [0, 1, 1, 0, 1, 0, 0, 1, 0, 0, 1, 0, 1, 1], [0, 1, 2, 3, 2, 1, 0]]
exact = []
for toy in (Params(n=12, log_n_poly=10, p_msg=7, sigma_lwe=1 << 8, sigma_glwe=1 << 8),
            Params(n=12, log_n_poly=11, l_bsk=1, beta_bsk=20, p_msg=7, sigma_lwe=1 << 8, sigma_glwe=4, bsk_group=2),
            Params(n=12, log_n_poly=11, l_bsk=2, beta_bsk=10, p_msg=7, sigma_lwe=1 << 8, sigma_glwe=4, bsk_group=2)):
    ctx, o = Context(toy, seed=4), orc.Oracle(toy, seed=4)
    msgs = np.arange(23) % 7
    ids = (np.arange(23) % 3).astype(np.uint32)
    cts = ctx.encrypt(msgs, 3)
    cts[22, :-1] = 0
    exact.append(bool(np.array_equal(ctx.bootstrap_batch(ctx.tvset(tabs), cts, ids), o.bootstrap_batch(cts, tabs, ids)[0])))
    ctx.close()
line = "%-40s exact=%s" % (os.environ.get("FBS_LIB", "in-tree").split("/")[-1], exact)
rng = np.random.default_rng(42)
for label, prm, sizes in (("P1024", P1024, (64, 256)), ("p15 k=1", choose_params(15, 70), (64, 256)), ("p31", choose_params(31, 325), (64, 256, 1024))):
    ctx = Context(prm, seed=1)
    p = prm.p_msg
    tables = [[0] + [int(v) for v in rng.integers(0, 2, p - 1)] for _ in range(16)]
    tv = ctx.tvset(tables)
    line += " | " + label
    for B in sizes:
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
        line += " %d: %.3f%s" % (B, prof["blind_rotate"]["ms"] / steps, "" if ok else " WRONG")
    ctx.close()
print(line, flush=True)
