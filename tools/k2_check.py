"""GLWE dimension k = 2 at N = 1024 (k_blind_rotate_pairs_k2): word-for-word against the oracle at toy n (ragged batches, every
table mode, trivial ciphertexts), then timing + decrypt check at a 128-bit set.   python3 tools/k2_check.py [batch] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tfhe_fbs_map_amd import Context, Params
from tfhe_fbs_map_amd.params import margin_sigmas, security_bits
from oracle import tfhe_oracle as orc

bad = 0
toy = Params(n=16, log_n_poly=10, k=2, l_bsk=1, beta_bsk=21, t_ksk=8, gamma_ksk=2, p_msg=7, sigma_lwe=1 << 8, sigma_glwe=4, bsk_group=2)
ctx, o = Context(toy, seed=4), orc.Oracle(toy, seed=4)
tabs = [[0, 1, 1, 0, 1, 0, 0], [0, 1, 1, 0, 1, 0, 0, 1, 0, 0, 1, 0, 1, 1], [0, 1, 2, 3, 2, 1, 0], [1, 1, 1, 0, 1, 0, 0, 1, 1, 1]]
tv = ctx.tvset(tabs)
for B in (1, 2, 3, 4, 5, 7, 21, 64, 301, 515, 1029):
    msgs = np.arange(B) % 7
    ids = (np.arange(B) % 4).astype(np.uint32)
    msgs[ids == 1] = np.arange(B)[ids == 1] % 14
    msgs[ids == 3] = np.arange(B)[ids == 3] % 10
    cts = ctx.encrypt(msgs, 3)
    if B > 2:
        cts[B - 1, :-1] = 0          # a trivial ciphertext: every step of its rotation is skipped
    got = ctx.bootstrap_batch(tv, cts, ids)
    ref, _ = o.bootstrap_batch(cts, tabs, ids)
    same = np.array_equal(got, ref)
    bad += not same
    print("toy n=16 B=%4d  GPU == oracle: %s  (%s)" % (B, same, ",".join(k for k in ctx.profile_kernels() if "blind" in k) if hasattr(ctx, "profile_kernels") else ""), flush=True)
ctx.close()

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
prm = Params(n=760, log_n_poly=10, k=2, l_bsk=1, beta_bsk=21, t_ksk=8, gamma_ksk=2, p_msg=15, sigma_lwe=456472211, sigma_glwe=4, bsk_group=2)
print("128-bit set: security %.1f bits, margin %.2f sigma at norm2 70" % (security_bits(prm), margin_sigmas(prm, 70)))
ctx = Context(prm, seed=1)
rng = np.random.default_rng(42)
tables = [[0] + [int(v) for v in rng.integers(0, 2, 14)] for _ in range(16)]
tv = ctx.tvset(tables)
for Bq in (B, 64, 256, 384, 512, 768, 2048):
    msgs = rng.integers(0, 15, Bq)
    ids = (np.arange(Bq) % 16).astype(np.uint32)
    d_in = torch.from_numpy(ctx.encrypt(msgs, nonce0=0).view(np.int64)).cuda()
    d_ids = torch.from_numpy(ids.view(np.int32)).cuda()
    d_out = torch.empty_like(d_in)
    for _ in range(2):
        ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), Bq, d_out.data_ptr())
    ctx.sync()
    ctx.profile(True); ctx.profile_read()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), Bq, d_out.data_ptr())
    ctx.sync()
    dt = time.perf_counter() - t0
    prof = ctx.profile_read()
    ok = np.array_equal(ctx.decrypt(d_out.cpu().numpy().view(np.uint64)), [tables[i][m] for i, m in zip(ids, msgs)])
    bad += not ok
    print("n=%d N=%d k=%d B=%5d  %.0f FBS/s  br %.3f ms (%s)  ks %.3f ms  ok=%s" % (
        prm.n, prm.N, prm.k, Bq, Bq * steps / dt, prof["blind_rotate"]["ms"] / steps, prof["blind_rotate"]["kernel"], prof["keyswitch"]["ms"] / steps, ok), flush=True)
# a few ciphertexts of the big set word for word (the oracle takes ~0.2 s per bootstrap here)
o = orc.Oracle(prm, seed=1)
msgs = rng.integers(0, 15, 6)
ids = (np.arange(6) % 16).astype(np.uint32)
cts = ctx.encrypt(msgs, nonce0=5)
got = ctx.bootstrap_batch(tv, cts, ids)
ref, _ = o.bootstrap_batch(cts, tables, ids)
same = np.array_equal(got, ref)
bad += not same
print("n=760 six ciphertexts GPU == oracle: %s" % same)
sys.exit(1 if bad else 0)
