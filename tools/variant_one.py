"""Bench + parity spot check of the library named by $FBS_LIB (kernel-variant experiments)."""
import json
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from tfhe_fbs_map_amd import _native as nat  # noqa: E402
from oracle import tfhe_oracle as orc  # noqa: E402

prm = nat.Params(n=12, log_n_poly=10, p_msg=7, sigma_lwe=1 << 8, sigma_glwe=1 << 8)
ctx = nat.Context(prm, seed=4)
o = orc.Oracle(prm, seed=4)
tabs = [[0, 1, 1, 0, 1, 0, 0], [0, 1, 1, 0, 1, 0, 0, 1, 0, 0, 1, 0, 1, 1], [0, 1, 2, 3, 2, 1, 0]]
msgs = np.arange(21) % 7
ids = (np.arange(21) % 3).astype(np.uint32)
msgs[ids == 1] = np.arange(21)[ids == 1] % 14
cts = ctx.encrypt(msgs, 3)
got = ctx.bootstrap_batch(ctx.tvset(tabs), cts, ids)
ref, _ = o.bootstrap_batch(cts, tabs, ids)
out = subprocess.run([sys.executable, "bench.py", "--steps", "8", "--cpu-sample", "0"], capture_output=True, text=True).stdout
d = json.loads(out)
print("%-40s exact=%s  %.0f FBS/s  br=%.2f ms ks=%.2f ms ok=%s" % (
    os.environ.get("LIBNAME"), np.array_equal(got, ref), d["value"], d["roofline"]["avg_launch_ms"],
    d["roofline"]["keyswitch_avg_launch_ms"], d["decrypt_ok"]), flush=True)
