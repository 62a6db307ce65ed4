#!/bin/bash
# SQ counters of the blind-rotation kernel for an arbitrary command: bash tools/pmc_any.sh <tag> python3 tools/secure_bench.py 1024 3
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $OUT/p$i -- "$@" > $OUT/p$i.out 2> $OUT/p$i.err
done
python3 - <<PY
import glob, csv, collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/p*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "blind_rotate" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt","w") as o:
    for k,v in sorted(acc.items()):
        line="%-24s %.4g" % (k, sum(v)/len(v)); print(line); o.write(line+"\n")
PY
cat $OUT/p1.out
