"""N = 4096: the 128-bit set the selector returns for p = 31 at norm2 = 325 (6 sigma), flat batch; timing + decrypt check."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tfhe_fbs_map_amd import Context, choose_params
from tfhe_fbs_map_amd.params import REFERENCE_MARGIN, bootstrap_cost, margin_sigmas
for prm, label in ((choose_params(31, 325), "6 sigma"), (choose_params(31, 325, poly_sizes=(9, 10, 11), floor_margin=REFERENCE_MARGIN), "relaxed, N <= 2048")):
    ctx = Context(prm, seed=1)
    rng = np.random.default_rng(1)
    tables = [[0] + [int(v) for v in rng.integers(0, 2, 30)] for _ in range(16)]
    B = 1024
    msgs = rng.integers(0, 31, B); ids = (np.arange(B) % 16).astype(np.uint32)
    tv = ctx.tvset(tables)
    d_in = torch.from_numpy(ctx.encrypt(msgs).view(np.int64)).cuda(); d_ids = torch.from_numpy(ids.view(np.int32)).cuda(); d_out = torch.empty_like(d_in)
    for _ in range(2): ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), B, d_out.data_ptr())
    ctx.sync(); ctx.profile(True); ctx.profile_read()
    t0 = time.perf_counter()
    for _ in range(4): ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), B, d_out.data_ptr())
    ctx.sync(); dt = (time.perf_counter() - t0) / 4
    p = ctx.profile_read()
    ok = np.array_equal(ctx.decrypt(d_out.cpu().numpy().view(np.uint64)), [tables[i][m] for i, m in zip(ids, msgs)])
    print("p=31 norm2=325 %s: n=%d N=%d l=%d beta=%d t=%d gamma=%d margin %.2f sigma, modelled cost %.2f: %.0f FBS/s (br %.2f ms [%s], ks %.2f ms per 1024) decrypt_ok=%s" % (
        label, prm.n, prm.N, prm.l_bsk, prm.beta_bsk, prm.t_ksk, prm.gamma_ksk, margin_sigmas(prm, 325), bootstrap_cost(prm), B / dt,
        p["blind_rotate"]["ms"] / 4, p["blind_rotate"]["kernel"], p["keyswitch"]["ms"] / 4, ok), flush=True)
    ctx.close()
