#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_ks
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_ks/$(echo $c | cut -c1-12 | tr " " _) -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
done
python3 - <<PY
import glob, csv, collections
for f in sorted(glob.glob("gpurun_out/pmc_ks/*/*/*counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "keyswitch" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print("%-24s %.4g" % (k, sum(v)/len(v)))
PY
