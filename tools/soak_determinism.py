"""Race hunt for the whole-CU kernels: every launch shape that shares LDS words between phases (hand-over = inverse exchange =
re-deal in k_blind_rotate_cu_pairs; accumulator words = hand-over in k_blind_rotate_cu; exchange buffer = landing words of the LDS
atomics in the k = 2 kernels; landing words of the general-GLWE kernel taken in turns) is run REPS times on the same inputs at
full n, with every CU busy, and every run must give the first run's ciphertexts bit for bit; the first run is checked against the
expected cleartexts.  A race shows as a rare mismatch.   python3 tools/soak_determinism.py [reps = 40]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tfhe_fbs_map_amd import Context, P1024, choose_params

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
CASES = [("P1024", P1024, (64, 256, 300, 512)), ("p15 128-bit", choose_params(15, 70), (256, 200)),
         ("p31 128-bit", choose_params(31, 325), (256, 300, 1024)), ("p31 one key bit", choose_params(31, 325, groups=(1,)), (256,)),
         ("p4 128-bit", choose_params(4, 2), (256, 512)),
         # GLWE dimension 2: every wave of a bootstrap clears its exchange buffer, all three ADD their products into it with LDS
         # atomics and read the total back, between two barriers per step (csrc/fbs_blind_rotate_k2.hip) -- one, two and four
         # bootstraps per workgroup
         ("p15 128-bit k=2", choose_params(15, 70, glwe_dims=(1, 2)), (200, 256, 300, 512, 1024, 1124)),
         ("p4 128-bit k=2", choose_params(4, 2, glwe_dims=(1, 2)), (256, 1024)),
         # GLWE dimension 3 at N = 512 (k_blind_rotate_glwe, csrc/fbs_blind_rotate_glwe.hip): products ADDED into the other components'
         # landing words as they are made, two sets of landing words taken in turns, ONE barrier per step -- one, two and three
         # bootstraps per workgroup, a launch the launcher cuts
         ("p4 128-bit k=3", choose_params(4, 2, glwe_dims=(1, 2, 3)), (200, 256, 500, 768, 1024, 1536)),
         ("p7 128-bit k=3", choose_params(7, 10, glwe_dims=(1, 2, 3)), (256, 768))]
bad = 0
for label, prm, sizes in CASES:
    ctx = Context(prm, seed=5)
    p = prm.p_msg
    rng = np.random.default_rng(9)
    tables = [[0] + [int(v) for v in rng.integers(0, 2, p - 1)] for _ in range(8)]
    tv = ctx.tvset(tables)
    for B in sizes:
        msgs = rng.integers(0, p, B)
        ids = (np.arange(B) % 8).astype(np.uint32)
        d_in = torch.from_numpy(ctx.encrypt(msgs, nonce0=1).view(np.int64)).cuda()
        d_ids = torch.from_numpy(ids.view(np.int32)).cuda()
        outs = [torch.empty_like(d_in) for _ in range(REPS)]
        ctx.profile(True); ctx.profile_read(reset=True)
        for o in outs:
            ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), B, o.data_ptr())
        ctx.sync()
        kernels = [k for k in ctx.profile_kernels() if "blind_rotate" in k]
        ctx.profile(False)
        first = outs[0]
        same = all(bool(torch.equal(first, o)) for o in outs[1:])
        ok = np.array_equal(ctx.decrypt(first.cpu().numpy().view(np.uint64)), [tables[i][m] for i, m in zip(ids, msgs)])
        bad += (not same) + (not ok)
        print("%-16s B=%5d  %-50s  %d runs identical: %s   decrypts: %s" % (label, B, ",".join(kernels), REPS, same, ok), flush=True)
    ctx.close()
print("FAILED" if bad else "all deterministic")
sys.exit(1 if bad else 0)
