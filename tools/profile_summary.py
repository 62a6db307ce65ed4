"""Per-kernel summary of a tools/profile_round.sh directory: average duration from the kernel trace, mean of every PMC
counter per launch, and the derived figures bench.py's `roofline` reads (profiles/rNN/pmc_blind_rotate.json)."""
import collections
import csv
import glob
import json
import re
import sys

root = sys.argv[1]


def short(name):
    m = re.search(r"(k_[a-z_0-9]+)\s*(<[^>]*>)?", name)
    if not m:
        return name
    k = (m.group(1) + (m.group(2) or "")).replace(" ", "")
    # the kernel trace prints defaulted template arguments; the launcher's names (fbs_profile_kernel, fbs_kernel_catalog) leave
    # them out: k_blind_rotate<10,6,3,4,true> is k_blind_rotate<10,6,3,4>, k_blind_rotate_cu<10,3,2,false> is k_blind_rotate_cu<10,3,2>
    # and k_blind_rotate_cu<10,3,2,true> its `lean` variant
    if k.startswith("k_blind_rotate_cu<"):
        return k.replace(",false>", ">").replace(",true>", ",lean>")
    return k.replace(",true>", ">")


out = collections.defaultdict(dict)
for f in glob.glob(root + "/kt/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if k.startswith("k_"):
            out[k]["trace"] = dict(calls=int(r["Calls"]), avg_ms=float(r["AverageNs"]) / 1e6, min_ms=float(r["MinNs"]) / 1e6,
                                   max_ms=float(r["MaxNs"]) / 1e6, percent=float(r["Percentage"]))
for f in glob.glob(root + "/pmc_*/*/*counter_collection.csv"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k.startswith("k_"):
            acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        out[k].setdefault("pmc_per_launch", {})[c] = sum(v) / len(v)
for k, rec in out.items():
    pmc = rec.get("pmc_per_launch", {})
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        # counters are KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B)
        rec["hbm_bytes_per_launch"] = (2 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024
    if "SQ_INSTS_VALU" in pmc and "SQ_WAVES" in pmc:
        rec["valu_per_wave"] = pmc["SQ_INSTS_VALU"] / pmc["SQ_WAVES"]
print(json.dumps(out, indent=1, sort_keys=True))
