#!/bin/bash
# LDS counters of the blind-rotation kernel for the library in $FBS_LIB (default: the product build)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_lds
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_lds -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
python3 - <<PY
import glob, csv, collections
for f in sorted(glob.glob("gpurun_out/pmc_lds/*/*counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "blind_rotate" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print("%-24s %.4g" % (k, sum(v)/len(v)))
PY
