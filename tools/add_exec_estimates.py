"""The reference's experiments/add_exec_estimates.py for this executor: append `boot_cost` and `total_cost` to a table of mapped
circuits -- here the 210 fixtures the reference's own mappers produced (tests/golden), with the statistics the reference logs for
them (`LutExecEnv.stats()`: nb_bootstrap, norm2_linprod) -- where the reference calls its patched concrete-optimizer per
(precision, sq_norm2) (add_exec_estimates.py:9-16) and analyse_results.py multiplies.  Pure host arithmetic (params.exec_estimate).

    python3 tools/add_exec_estimates.py > profiles/r03/exec_estimates.csv
"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import fixture_names, load_fixture          # noqa: E402
from tfhe_fbs_map_amd.params import exec_estimate              # noqa: E402

w = csv.writer(sys.stdout)
w.writerow("bench mapper fbs_size nb_bootstrap norm2_linprod n N l beta t gamma key_bits_per_step margin_sigmas boot_cost total_cost "
           "est_ms_per_1000_samples".split())
seen = {}
for name in fixture_names():
    rec = load_fixture(name)
    if "__" not in name or "_p" not in name:
        continue
    bench, rest = name.split("__", 1)
    mapper, p = rest.rsplit("_p", 1)
    if not p.isdigit() or not rec.get("stats"):
        continue
    st = rec["stats"]
    key = (int(p), st["norm2_linprod"])
    try:
        est = seen[key] = seen.get(key) or exec_estimate(int(p), st["norm2_linprod"], 1)
    except ValueError:
        continue
    prm = est["params"]
    total = st["nb_bootstrap"] * est["boot_cost"]
    w.writerow([bench, mapper, p, st["nb_bootstrap"], st["norm2_linprod"], prm.n, prm.N, prm.l_bsk, prm.beta_bsk, prm.t_ksk, prm.gamma_ksk,
                prm.bsk_group, round(est["margin_sigmas"], 2), round(est["boot_cost"], 3), round(total, 2),
                round(total * 1000 / 107e3 * 1e3, 2)])
