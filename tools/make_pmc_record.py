"""Fold a tools/profile_round.sh summary into the record bench.py reads for `roofline` (profiles/rNN/pmc_blind_rotate.json):
    python3 tools/make_pmc_record.py <summary.json> <n> <key bits per step> <record.json> [units per launch = 1024]
Every blind-rotation kernel of the summary gets an entry keyed by its instantiation name."""
import json
import sys

summary, n, group, record = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
units = int(sys.argv[5]) if len(sys.argv) > 5 else 1024
data = json.load(open(summary))
try:
    out = json.load(open(record))
except FileNotFoundError:
    out = {}
for kernel, rec in data.items():
    pmc = rec.get("pmc_per_launch", {})
    if "blind_rotate" not in kernel or "SQ_INSTS_VALU" not in pmc:
        continue
    steps = n / group
    out[kernel] = {
        "units_per_launch": units, "n": n, "steps_per_bootstrap": steps,
        "SQ_INSTS_VALU_per_launch": pmc["SQ_INSTS_VALU"], "SQ_WAVES": pmc.get("SQ_WAVES"),
        "valu_per_wave_per_step": round(pmc["SQ_INSTS_VALU"] / pmc["SQ_WAVES"] / steps, 1) if pmc.get("SQ_WAVES") else None,
        "hbm_bytes_per_launch": rec.get("hbm_bytes_per_launch"),
        "rocprof_avg_launch_ms": rec.get("trace", {}).get("avg_ms"), "rocprof_min_launch_ms": rec.get("trace", {}).get("min_ms"),
        "GRBM_GUI_ACTIVE": pmc.get("GRBM_GUI_ACTIVE"),
        "source": "%s (tools/profile_round.sh: rocprofv3 --pmc, one pass per counter group; FETCH_SIZE x2 + WRITE_SIZE, KB -> bytes)" % summary,
    }
json.dump(out, open(record, "w"), indent=1)
print(json.dumps({k: v["valu_per_wave_per_step"] for k, v in out.items()}))
