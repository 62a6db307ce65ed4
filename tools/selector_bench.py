"""The 128-bit sets the selector returns for a few (p, norm2), each on a flat batch of 1024 bootstraps: what was chosen,
timing, and a decryption check of every output.    python3 tools/selector_bench.py [p:norm2 ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                                          # noqa: E402
import torch                                                                # noqa: E402
from tfhe_fbs_map_amd import Context, choose_params, security_bits          # noqa: E402
from tfhe_fbs_map_amd.params import bootstrap_cost, margin_sigmas           # noqa: E402

cases = [tuple(float(x) for x in a.split(":")) for a in sys.argv[1:]] or [(4, 2), (15, 70), (15, 281), (31, 325), (63, 100)]
for p, norm2 in cases:
    p = int(p)
    prm = choose_params(p, norm2)
    ctx = Context(prm, seed=1)
    rng = np.random.default_rng(1)
    tables = [[0] + [int(v) for v in rng.integers(0, 2, p - 1)] for _ in range(16)]
    B = int(os.environ.get("BATCH", 1024))
    msgs = rng.integers(0, p, B)
    ids = (np.arange(B) % 16).astype(np.uint32)
    tv = ctx.tvset(tables)
    d_in = torch.from_numpy(ctx.encrypt(msgs).view(np.int64)).cuda()
    d_ids = torch.from_numpy(ids.view(np.int32)).cuda()
    d_out = torch.empty_like(d_in)
    for _ in range(2):
        ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), B, d_out.data_ptr())
    ctx.sync()
    ctx.profile(True)
    ctx.profile_read()
    t0 = time.perf_counter()
    steps = 4
    for _ in range(steps):
        ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), B, d_out.data_ptr())
    ctx.sync()
    dt = (time.perf_counter() - t0) / steps
    prof = ctx.profile_read()
    got = ctx.decrypt(d_out.cpu().numpy().view(np.uint64))
    want = np.array([tables[i][m] for i, m in zip(ids, msgs)])
    print(json.dumps({
        "p": p, "norm2": norm2,
        "params": dict(n=prm.n, N=prm.N, l=prm.l_bsk, beta=prm.beta_bsk, t=prm.t_ksk, gamma=prm.gamma_ksk, bsk_group=prm.bsk_group),
        "security_bits": round(security_bits(prm), 1), "margin_sigmas": round(margin_sigmas(prm, norm2), 2),
        "model_cost": round(bootstrap_cost(prm), 3), "fbs_per_s": round(B / dt), "ms_per_1024": round(dt * 1e3, 3),
        "keyswitch_ms": round(prof["keyswitch"]["ms"] / steps, 3), "blind_rotate_ms": round(prof["blind_rotate"]["ms"] / steps, 3),
        "kernels": [prof["keyswitch"]["kernel"], prof["blind_rotate"]["kernel"]], "all_decrypt_correct": bool(np.array_equal(got, want))}))
    ctx.close()
