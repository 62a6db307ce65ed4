"""Every golden fixture (programs written by the reference's mappers, T samples of its harness inputs) evaluated through
`LutExecEnv.eval` with the DEFAULT ExecConfig: 128-bit parameters from params.choose_params at each program's own
(p, norm2_linprod), fresh keys.  One-off validation; the summary goes to profiles/."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.helpers import fixture_names, load_fixture, subsample
from tfhe_fbs_map_amd import ExecConfig, parse_fbs
from tfhe_fbs_map_amd.params import margin_sigmas, security_bits

T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
skip_big = ("trivium_stream", "kreyvium_stream")
cfg = ExecConfig(seed=2026, reduced_noise=bool(os.environ.get("REDUCED_NOISE")))   # REDUCED_NOISE=1: the benchmark sets (whole-CU kernels)
rows, bad = [], []
t_all = time.time()
for name in fixture_names():
    if name.startswith(skip_big):
        continue
    rec = load_fixture(name)
    ins, expect = subsample(rec, T)
    env = parse_fbs(rec["fbs"], inputs=rec["program_inputs"], merge_linear_prods=name != "edge_nomerge")
    t0 = time.time()
    try:
        got = env.eval(ins, config=cfg)
    except Exception as e:      # noqa: BLE001
        bad.append((name, repr(e)))
        continue
    ok = set(got) == set(expect) and all(
        (int(got[k]) == int(v)) if isinstance(v, int) else np.array_equal(np.asarray(got[k]).reshape(-1), v) for k, v in expect.items())
    st = env.stats()
    prog = next(reversed(cfg._programs.values()))[0] if cfg._programs else None      # the program this eval loaded or reused
    rows.append(dict(name=name, ok=bool(ok), nb_bootstrap=st["nb_bootstrap"], norm2=st["norm2_linprod"], seconds=round(time.time() - t0, 2),
                     shared_rotations=bool(prog and prog.fused), rotations=prog.n_rotations if prog else None,
                     N=prog.ctx.params.N if prog else None, k=prog.ctx.params.k if prog else None))
    if not ok:
        bad.append((name, "wrong output"))
    if len(cfg._contexts) > 6:                     # keys are large: keep a few contexts
        for key in list(cfg._contexts)[:-3]:
            cfg._contexts.pop(key).close()
        cfg._programs.clear()
    print("%-44s %s  %5d FBS  norm2 %4d  %.1fs" % (name, "ok " if ok else "BAD", st["nb_bootstrap"], st["norm2_linprod"], time.time() - t0), flush=True)
sets = {}
summary = dict(samples=T, fixtures=len(rows), all_ok=not bad, failures=bad, bootstraps=sum(r["nb_bootstrap"] for r in rows) * T,
               rotations=sum((r["rotations"] or 0) for r in rows) * T, programs_with_shared_rotations=sum(r["shared_rotations"] for r in rows),
               poly_sizes=sorted({r["N"] for r in rows if r["N"]}), programs_at_glwe_dimension_2=sum(1 for r in rows if r.get("k") == 2), programs_at_glwe_dimension_3=sum(1 for r in rows if r.get("k") == 3),
               seconds=round(time.time() - t_all, 1), rows=rows)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(summary, open("gpurun_out/%s_all_fixtures.json" % ("reduced_noise" if os.environ.get("REDUCED_NOISE") else "secure"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "rows"}))
