"""What the general-GLWE kernel costs per modelled instruction, shape by shape: for every (N, k, key bits per step) it is built for, the
cheapest 128-bit set for (p, norm2) = (4, 2) of that shape, timed at two full rounds of workgroups -- the figures behind
params.GLWE_US_PER_MINSTR.      python3 tools/glwe_calibrate.py [steps = 3] > profiles/r04/glwe_calibration.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.glwe_candidates import best_for, instr, time_set

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
print("# python3 tools/glwe_calibrate.py %d: ms per launch (key switch + blind rotation) of two full rounds of workgroups, per shape of k_blind_rotate_glwe" % steps)
print("# instr = modelled FP64 wave-instructions per bootstrap (tools/glwe_candidates.instr); us/Minstr = ms per 1024 bootstraps / instr")
for log_n in (8, 9, 10):
    for k in (2, 3, 4):
        if log_n == 10 and k == 4:
            continue
        fpw = 2 if log_n == 10 else 12 // (k + 1)
        B = 2 * 256 * fpw
        for group in (1, 2):
            b = best_for(4, 2, k, log_n, groups=(group,))
            if b is None:
                print("N=%d k=%d bits/step=%d: no 128-bit set reaches 6 sigma at p = 4" % (1 << log_n, k, group), flush=True)
                continue
            prm = b[1]
            ms, kern, ok = time_set(prm, B, steps)
            ins = instr(prm.n, prm.N, prm.log_n_poly, prm.k, prm.l_bsk, prm.bsk_group)
            print("N=%d k=%d bits/step=%d: n=%d l=%d beta=%d t=%d g=%d  instr %.3f M  B=%d %.3f ms  -> %.1f k FBS/s, %.2f us/Minstr  %s%s" % (
                prm.N, k, group, prm.n, prm.l_bsk, prm.beta_bsk, prm.t_ksk, prm.gamma_ksk, ins / 1e6, B, ms, B / ms, ms * 1024.0 / B / (ins / 1e6) , kern, "" if ok else " WRONG"), flush=True)
