#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats of bench.py, then HBM traffic counters in their own passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python3 bench.py --steps 10 --cpu-sample 0 > gpurun_out/bench_kt.json 2> gpurun_out/bench_kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
python3 - <<PY
import glob, csv, collections, json
out={}
for tag in ("fetch","write"):
    for f in glob.glob("gpurun_out/prof_%s/*/*counter_collection.csv"%tag):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k="blind_rotate" if "blind_rotate" in r["Kernel_Name"] else "keyswitch" if "keyswitch" in r["Kernel_Name"] else None
            if k: acc[(k,r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k,c),v in acc.items(): out["%s.%s"%(k,c)]=sum(v)/len(v)
print(json.dumps(out))
json.dump(out, open("gpurun_out/traffic_raw.json","w"))
PY
cat gpurun_out/prof_kt/*/*kernel_stats.csv | head -6
cat gpurun_out/bench_kt.json
