"""Launch-time staircase of k_blind_rotate_glwe at the 128-bit set for p = 4 of a shape (default k = 3, N = 512, two key bits per step).
    python3 tools/glwe_staircase.py [k = 3] [log_n = 9] [steps = 4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.glwe_candidates import best_for, time_set
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 9
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
for p, norm2 in ((4, 2), (7, 10)):
    prm = best_for(p, norm2, k, log_n)[1]
    print("== p=%d norm2=%g: n=%d N=%d k=%d l=%d beta=%d bits/step=%d t=%d g=%d" % (p, norm2, prm.n, prm.N, prm.k, prm.l_bsk, prm.beta_bsk, prm.bsk_group, prm.t_ksk, prm.gamma_ksk), flush=True)
    for B in (1, 64, 128, 256, 384, 512, 768, 769, 1024, 1536, 1537, 2304, 3072, 6144):
        ms, kern, ok = time_set(prm, B, steps)
        print("  B=%5d  %7.3f ms  %6.1f k FBS/s  %s%s" % (B, ms, B / ms, kern, "" if ok else " WRONG"), flush=True)
