"""The launch-time staircase of every kernel family schedule.LAUNCH_FAMILIES prices layouts on: one key switch + blind rotation
launch of B bootstraps, B = 1 .. 2048, at the set each family was named after.
    python3 tools/launch_staircases.py [steps = 5] > profiles/r04/launch_staircases.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tfhe_fbs_map_amd import Context, P1024, choose_params
from tfhe_fbs_map_amd.schedule import launch_family, launch_ms

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
SIZES = (1, 64, 128, 256, 257, 384, 512, 513, 640, 768, 769, 896, 1024, 1124, 1280, 1536, 2048, 2304, 3072)
SETS = (("p1024", P1024), ("k2", choose_params(15, 70, glwe_dims=(1, 2))), ("k2 (p = 4)", choose_params(4, 2, glwe_dims=(1, 2))),
        ("k3", choose_params(4, 2, glwe_dims=(1, 2, 3))), ("k3 (p = 7)", choose_params(7, 10, glwe_dims=(1, 2, 3))),
        ("n2048", choose_params(15, 70)), ("n2048_l2", choose_params(31, 325)))
print("# python3 tools/launch_staircases.py %d  on %s" % (steps, torch.cuda.get_device_name(0)))
print("# per launch of B bootstraps: ms measured (key switch + blind rotation, HIP events in the library) | modelled by schedule.launch_ms | kernels")
for label, prm in SETS:
    ctx = Context(prm, seed=1)
    p = prm.p_msg
    rng = np.random.default_rng(42)
    tables = [[0] + [int(v) for v in rng.integers(0, 2, p - 1)] for _ in range(16)]
    tv = ctx.tvset(tables)
    print("== %s: n=%d N=%d k=%d l=%d key bits per step %d -> family %s" % (label, prm.n, prm.N, prm.k, prm.l_bsk, prm.bsk_group, launch_family(prm)), flush=True)
    for B in SIZES:
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
        kernels = [k for k in ctx.profile_kernels() if "blind_rotate" in k]
        prof = ctx.profile_read()
        ctx.profile(False)
        ok = bool(np.array_equal(ctx.decrypt(d_out.cpu().numpy().view(np.uint64)), [tables[i][m] for i, m in zip(ids, msgs)]))
        ms = (prof["blind_rotate"]["ms"] + prof["keyswitch"]["ms"]) / steps
        print("  B=%5d  %7.3f ms | model %7.3f | %s%s" % (B, ms, launch_ms(B, params=prm), " + ".join(kernels), "" if ok else "  WRONG"), flush=True)
    ctx.close()
