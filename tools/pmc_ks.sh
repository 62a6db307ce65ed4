#!/bin/bash
# fabric traffic of the key-switch kernel (FETCH_SIZE / WRITE_SIZE, separate passes) for the grid order in $FBS_KS_TILES_MAJOR
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_ks_${1:-default}
rm -rf $OUT && mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-secure > /dev/null 2> $OUT/$c.err
done
python3 - <<PY
import glob, csv, collections, json
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "keyswitch" in r["Kernel_Name"] or "blind_rotate" in r["Kernel_Name"]:
            acc[("keyswitch" if "keyswitch" in r["Kernel_Name"] else "blind_rotate", r["Counter_Name"])].append(float(r["Counter_Value"]))
out={"%s.%s_KB"%k: sum(v)/len(v) for k,v in acc.items()}
for k in ("keyswitch","blind_rotate"):
    out[k+".bytes_per_launch"]=(2*out.get(k+".FETCH_SIZE_KB",0)+out.get(k+".WRITE_SIZE_KB",0))*1024
print(json.dumps(out)); json.dump(out, open("$OUT/summary.json","w"))
PY
