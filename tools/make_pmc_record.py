"""Fold a tools/profile_round.sh summary into the record bench.py reads for `roofline` (profiles/rNN/pmc_blind_rotate.json):
    python3 tools/make_pmc_record.py <summary.json> <n> <key bits per step> <record.json> [units per launch = 1024] [tracked source name]
Every blind-rotation kernel of the summary gets an entry keyed by (instantiation name, n, units per launch), stamped with the SHA-256 of the
kernel sources it was collected on (bench.py refuses a record taken on other sources)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_sources_sha256, pmc_key  # noqa: E402

summary, n, group, record = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
units = int(sys.argv[5]) if len(sys.argv) > 5 else 1024
tracked = sys.argv[6] if len(sys.argv) > 6 else summary
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
    entry = {
        "units_per_launch": units, "n": n, "steps_per_bootstrap": steps, "csrc_sha256": kernel_sources_sha256(),
        "SQ_INSTS_VALU_per_launch": pmc["SQ_INSTS_VALU"], "SQ_WAVES": pmc.get("SQ_WAVES"),
        "valu_per_wave_per_step": round(pmc["SQ_INSTS_VALU"] / pmc["SQ_WAVES"] / steps, 1) if pmc.get("SQ_WAVES") else None,
        "hbm_bytes_per_launch": rec.get("hbm_bytes_per_launch"),
        "rocprof_avg_launch_ms": rec.get("trace", {}).get("avg_ms"), "rocprof_min_launch_ms": rec.get("trace", {}).get("min_ms"),
        "GRBM_GUI_ACTIVE": pmc.get("GRBM_GUI_ACTIVE"),
        "source": "%s (tools/profile_round.sh: rocprofv3 --pmc, one pass per counter group; FETCH_SIZE x2 + WRITE_SIZE, KB -> bytes)" % tracked,
    }
    if "SQ_INSTS_VALU_INT32" in pmc:
        # instruction mix: everything but the 32-bit integer class is charged a full 4-cycle issue slot (FP64 arithmetic,
        # conversions, 64-bit integer); the 32-bit ones 2 (MI355X_MICROARCH.md, v_fma_f32 at two waves per SIMD)
        entry["valu_mix_per_launch"] = {c: pmc[c] for c in pmc if c.startswith("SQ_INSTS_VALU_")}
        entry["valu_64bit_per_launch"] = pmc["SQ_INSTS_VALU"] - pmc["SQ_INSTS_VALU_INT32"]
    # one kernel instantiation runs at several parameter sets and launch sizes (k_blind_rotate_pairs_k2<10,4>: the p = 15 set at
    # n = 734 and the p = 4 set at n = 630): the key names all three, bench.pmc_key()
    out[pmc_key(kernel, n, units)] = entry
json.dump(out, open(record, "w"), indent=1)
print(json.dumps({k: v["valu_per_wave_per_step"] for k, v in out.items()}))
