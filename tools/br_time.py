"""Per-launch time of the blind rotation of the library named by $FBS_LIB at full batches of each shipped kernel family (kernel-variant
experiments, tools/build_variants.sh; results are checked by decryption and a variant that computes something else says WRONG).
    python3 tools/br_time.py [steps = 6] [batch = 1024]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tfhe_fbs_map_amd import Context, P1024, choose_params

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
line = "%-36s" % os.environ.get("FBS_LIB", "in-tree").split("/")[-1]
rng = np.random.default_rng(42)
for label, prm in (("P1024", P1024), ("p15 k=2", choose_params(15, 70, glwe_dims=(1, 2))), ("p15 k=1", choose_params(15, 70)),
                   ("p4 k=1", choose_params(4, 2)), ("p31", choose_params(31, 325))):
    ctx = Context(prm, seed=1)
    p = prm.p_msg
    tables = [[0] + [int(v) for v in rng.integers(0, 2, p - 1)] for _ in range(16)]
    tv = ctx.tvset(tables)
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
    line += " | %s %.3f ms%s (%s)" % (label, prof["blind_rotate"]["ms"] / steps, "" if ok else " WRONG", prof["blind_rotate"]["kernel"].replace("k_blind_rotate", ""))
    ctx.close()
print(line, flush=True)
