"""The launch shapes of GLWE dimension k = 2 side by side (csrc/fbs_blind_rotate_k2.hip): three waves per bootstrap
(k_blind_rotate_pairs_k2<10, 1 | 2 | 4>) against ONE bootstrap on the twelve waves of a workgroup (k_blind_rotate_cu_k2).
First every shape word for word against the oracle at toy n (ragged batches, every table mode, trivial and maximal ciphertexts),
then per-launch times at the 128-bit set the selector ships for (15, 70) with the outputs decrypted and a few ciphertexts held to
the oracle.      python3 tools/k2_shapes.py [steps = 5] [sizes ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tfhe_fbs_map_amd import Context, Params, choose_params
from oracle import tfhe_oracle as orc

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
sizes = [int(v) for v in sys.argv[2:]] or [64, 128, 256, 300, 384, 512, 768, 1024]
SHAPES = {3: "three waves per bootstrap", 12: "twelve waves per bootstrap"}
bad = 0
tabs = [[0, 1, 1, 0, 1, 0, 0], [0, 1, 1, 0, 1, 0, 0, 1, 0, 0, 1, 0, 1, 1], [0, 1, 2, 3, 2, 1, 0], [1, 1, 1, 0, 1, 0, 0, 1, 1, 1]]
for beta in (21, 17):
    toy = Params(n=16, log_n_poly=10, k=2, l_bsk=1, beta_bsk=beta, t_ksk=8, gamma_ksk=2, p_msg=7, sigma_lwe=1 << 8, sigma_glwe=4, bsk_group=2)
    ctx, o = Context(toy, seed=4), orc.Oracle(toy, seed=4)
    tv = ctx.tvset(tabs)
    for B in (1, 5, 64, 301):
        msgs = np.arange(B) % 7
        ids = (np.arange(B) % 4).astype(np.uint32)
        msgs[ids == 1] = np.arange(B)[ids == 1] % 14
        msgs[ids == 3] = np.arange(B)[ids == 3] % 10
        cts = ctx.encrypt(msgs, 3)
        if B > 2:
            cts[B - 1, :-1] = 0
            cts[B // 2, :] = orc.Q - 1
        ref, _ = o.bootstrap_batch(cts, tabs, ids)
        for shape in SHAPES:
            ctx.tune(br_k2_shape=shape)
            ctx.profile(True); ctx.profile_read(reset=True)
            got = ctx.bootstrap_batch(tv, cts, ids)
            same = bool(np.array_equal(got, ref))
            bad += not same
            print("toy beta=%d B=%4d shape %2d  GPU == oracle: %s  (%s)" % (beta, B, shape, same, ",".join(k for k in ctx.profile_kernels() if "blind" in k)), flush=True)
    ctx.close()
if bad:
    print("MISMATCH at toy size: no timing")
    sys.exit(1)

prm = choose_params(15, 70, glwe_dims=(1, 2))
assert prm.k == 2
ctx = Context(prm, seed=1)
rng = np.random.default_rng(42)
tables = [[0] + [int(v) for v in rng.integers(0, 2, 14)] for _ in range(16)]
tv = ctx.tvset(tables)
for B in sizes:
    msgs = rng.integers(0, 15, B)
    ids = (np.arange(B) % 16).astype(np.uint32)
    d_in = torch.from_numpy(ctx.encrypt(msgs, nonce0=0).view(np.int64)).cuda()
    d_ids = torch.from_numpy(ids.view(np.int32)).cuda()
    d_out = torch.empty_like(d_in)
    line = "n=%d B=%5d " % (prm.n, B)
    for shape in SHAPES:
        ctx.tune(br_k2_shape=shape)
        for _ in range(2):
            ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), B, d_out.data_ptr())
        ctx.sync()
        ctx.profile(True); ctx.profile_read(reset=True)
        for _ in range(steps):
            ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), B, d_out.data_ptr())
        ctx.sync()
        prof = ctx.profile_read()
        ok = bool(np.array_equal(ctx.decrypt(d_out.cpu().numpy().view(np.uint64)), [tables[i][m] for i, m in zip(ids, msgs)]))
        bad += not ok
        line += " | %-30s %6.3f ms ok=%s" % (prof["blind_rotate"]["kernel"], prof["blind_rotate"]["ms"] / steps, ok)
    print(line, flush=True)
o = orc.Oracle(prm, seed=1)
msgs = rng.integers(0, 15, 6)
ids = (np.arange(6) % 16).astype(np.uint32)
cts = ctx.encrypt(msgs, nonce0=5)
ref, _ = o.bootstrap_batch(cts, tables, ids)
for shape in SHAPES:
    ctx.tune(br_k2_shape=shape)
    same = bool(np.array_equal(ctx.bootstrap_batch(tv, cts, ids), ref))
    bad += not same
    print("n=%d six ciphertexts, shape %2d, GPU == oracle: %s" % (prm.n, shape, same))
sys.exit(1 if bad else 0)
