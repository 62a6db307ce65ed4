"""Per-launch time of k_blind_rotate_glwe of the library named by $FBS_LIB at 128-bit sets of a few (k, N) (kernel-variant experiments).
    python3 tools/glwe_time.py [steps = 4] [batches ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfhe_fbs_map_amd import Params
from tools.glwe_candidates import best_for, time_set

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
batches = [int(v) for v in sys.argv[2:]] or [768, 1024]
line = "%-30s" % os.environ.get("FBS_LIB", "in-tree").split("/")[-1]
for p, norm2, k, log_n in ((4, 2, 3, 9), (4, 2, 2, 9), (4, 2, 4, 8), (7, 10, 3, 9), (7, 10, 3, 10)):
    prm = best_for(p, norm2, k, log_n)[1]
    line += " | p=%d k=%d N=%d l=%d g=%d n=%d:" % (p, k, prm.N, prm.l_bsk, prm.bsk_group, prm.n)
    for B in batches:
        ms, kern, ok = time_set(prm, B, steps)
        line += " %d: %.2f%s" % (B, ms, "" if ok else " WRONG")
print(line, flush=True)
