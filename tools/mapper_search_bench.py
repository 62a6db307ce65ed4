"""SURVEY 8(f)4 measurement: the coefficient search of the reference mapper (map_to_fbs.py:363-392) as one kernel launch
over all candidates, against the numpy restatement of the reference's loop (oracle/mapper_search_oracle.py, the CPU
baseline: one thread, as the reference runs it) on the same inputs.  Recorded calls of the largest sizes plus synthetic
cones at the reference's limit of 16 support variables (65 536 rows)."""
import gzip, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import mapper_search_oracle as mso
from tfhe_fbs_map_amd.mapper_search import CoefSearcher

s = CoefSearcher(0)
cases = json.loads(gzip.open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "_mapper_search.json.gz")).read())["cases"]
work = [(np.array(c["x"]), np.array(c["y"]), np.array(c["tt"]), c["fbs_size"], c["max_fbs_size"], "recorded") for c in cases if len(c["tt"]) >= 1024]
rng = np.random.default_rng(7)
for R, sx, sy, p in ((1 << 14, 9, 12, 15), (1 << 16, 14, 15, 15), (1 << 16, 15, 15, 31), (1 << 16, 30, 29, 31)):
    x, y = rng.integers(0, sx + 1, R), rng.integers(0, sy + 1, R)
    v = 2 * x - y
    work.append((x, y, ((v - v.min()) % 2).astype(np.int64), p, 2 * p, "synthetic, solvable"))
    work.append((x, y, rng.integers(0, 2, R), p, 2 * p, "synthetic, unsolvable"))
s.search(np.column_stack([work[0][0], work[0][1]]), work[0][2], work[0][3], work[0][4])     # warm-up
rows = []
for x, y, tt, p, maxp, kind in work:
    n_cand = (2 * min(mso.mvt_size(x), mso.mvt_size(y)) + 1) * (max(mso.mvt_size(x), mso.mvt_size(y)) + 1)
    t0 = time.perf_counter(); want = mso.find_lincomb_coefs_search(x, y, tt, p, maxp); t_cpu = time.perf_counter() - t0
    xy = np.column_stack([x, y])
    t0 = time.perf_counter(); got = s.search(xy, tt, p, maxp); t_gpu = time.perf_counter() - t0
    same = (got[0] is None and want[0] is None) or (got[0] is not None and want[0] is not None and tuple(got[0]) == tuple(want[0]) and np.array_equal(got[1], want[1]))
    # bytes the kernel reads: two passes over x, y (4 B each) and one over tt, per candidate that survives the first pass at most
    rows.append(dict(rows=len(tt), candidates=n_cand, fbs_size=p, kind=kind, found=got[0] is not None, same_as_cpu=bool(same),
                     cpu_ms=round(t_cpu * 1e3, 2), gpu_call_ms=round(t_gpu * 1e3, 3), kernel_ms=round(s.last_kernel_ms, 4),
                     read_GBps_upper=round(n_cand * len(tt) * 17 / max(s.last_kernel_ms, 1e-6) / 1e6, 1)))
    print(rows[-1], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/mapper_search_bench.json", "w"), indent=1)
