"""Two key bits per blind-rotation step (bsk_group = 2): parity against the oracle on toy sizes, then timing of the
128-bit set and of P1024 with and without it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import tfhe_oracle as orc
from tfhe_fbs_map_amd import Context, Params, P1024, choose_params

tables = [[0, 1, 1, 0, 1, 0, 0], [0, 1, 2, 3, 2, 1, 0], [0, 1, 1, 0, 1, 0, 0, 1, 0, 0, 1, 0, 1, 1]]
msgs = np.concatenate([np.arange(len(t)) for t in tables])
ids = np.concatenate([np.full(len(t), i) for i, t in enumerate(tables)]).astype(np.uint32)
for log_n in (10, 11):
    for l, beta in ((1, 20), (3, 7), (2, 10)):
        prm = Params(n=12, log_n_poly=log_n, l_bsk=l, beta_bsk=beta, p_msg=7, sigma_lwe=1 << 8, sigma_glwe=1 << 8, bsk_group=2)
        ctx, o = Context(prm, seed=4), orc.Oracle(prm, seed=4)
        keys_equal = all(np.array_equal(v, o.keys()[k]) for k, v in ctx.export_keys().items())
        cts = ctx.encrypt(msgs, nonce0=11)
        got = ctx.bootstrap_batch(ctx.tvset(tables), cts, ids)
        ref, _ = o.bootstrap_batch(cts, tables, ids)
        print("N=%d l=%d beta=%d keys=%s exact=%s decrypt=%s" % (1 << log_n, l, beta, keys_equal, np.array_equal(got, ref),
              np.array_equal(ctx.decrypt(got), np.concatenate([np.array(t) for t in tables]))), flush=True)
        ctx.close()


def timeit(prm, B=1024, steps=5, label=""):
    ctx = Context(prm, seed=1)
    rng = np.random.default_rng(42)
    p = prm.p_msg
    tabs = [[0] + [int(v) for v in rng.integers(0, 2, p - 1)] for _ in range(16)]
    tv = ctx.tvset(tabs)
    m = rng.integers(0, p, B)
    idv = (np.arange(B) % 16).astype(np.uint32)
    d_in = torch.from_numpy(ctx.encrypt(m, nonce0=0).view(np.int64)).cuda()
    d_ids = torch.from_numpy(idv.view(np.int32)).cuda()
    d_out = torch.empty_like(d_in)
    for _ in range(2):
        ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), B, d_out.data_ptr())
    ctx.sync(); ctx.profile(True); ctx.profile_read()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), B, d_out.data_ptr())
    ctx.sync(); dt = time.perf_counter() - t0
    prof = ctx.profile_read()
    out = d_out.cpu().numpy().view(np.uint64)
    ok = np.array_equal(ctx.decrypt(out), [tabs[i][x] for i, x in zip(idv, m)])
    print("%-28s n=%d N=%d l=%d beta=%d group=%d: %.0f FBS/s  br %.3f ms (%s) ks %.3f ms ok=%s" % (
        label, prm.n, prm.N, prm.l_bsk, prm.beta_bsk, prm.bsk_group, B * steps / dt, prof["blind_rotate"]["ms"] / steps,
        prof["blind_rotate"]["kernel"], prof["keyswitch"]["ms"] / steps, ok), flush=True)
    ctx.close()


sec = choose_params(15, 70)
timeit(sec, label="secure, one bit per step")
timeit(sec.replace(bsk_group=2), label="secure, two bits per step")
timeit(P1024, label="P1024, one bit per step")
timeit(P1024.replace(bsk_group=2), label="P1024, two bits per step")
