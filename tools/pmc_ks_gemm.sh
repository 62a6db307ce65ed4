#!/bin/bash
# Matrix-core counters of the key-switch GEMM (k_ks_gemm) for a command: bash tools/pmc_ks_gemm.sh <tag> python3 tools/secure_bench.py 1024 5
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_ksgemm_$TAG
rm -rf $OUT && mkdir -p $OUT
i=0
for c in "SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/p$i -- "$@" > $OUT/p$i.out 2> $OUT/p$i.err
done
python3 - <<PY
import glob, csv, collections, json
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/p*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        for key in ("k_ks_gemm<", "k_ks_gemm_finish", "k_ks_digits"):
            if key in n: acc[key.rstrip("<")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out={k:{c:sum(v)/len(v) for c,v in d.items()} for k,d in acc.items()}
json.dump(out, open("$OUT/summary.json","w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
