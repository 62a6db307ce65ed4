#!/bin/bash
# A/B of kernel-variant libraries in gpurun_exp/ against the in-tree one, on one GPU: an oracle spot check, then the launches the
# profile sets time (headline, one bootstrap per CU, the 128-bit sets).  usage (GPU box): bash tools/ab_variants.sh > gpurun_out/ab.txt
for lib in tfhe_fbs_map_amd/libfbsexec.so gpurun_exp/*.so; do
  echo "== $lib"
  export FBS_LIB=$PWD/$lib LIBNAME=$lib
  timeout -k 10 200 python3 tools/variant_one.py || exit 1
  timeout -k 10 120 python3 tools/secure_bench.py 1024 5 15 70 || exit 1
  timeout -k 10 120 python3 tools/secure_bench.py 256 8 15 70 || exit 1
  timeout -k 10 120 python3 tools/secure_bench.py 1024 4 31 325 || exit 1
  timeout -k 10 120 python3 tools/secure_bench.py 1024 5 4 2 || exit 1
  timeout -k 10 120 python3 tools/secure_bench.py 1024 5 15 70 1 || exit 1
  for b in 256 512; do
    timeout -k 10 120 python3 bench.py --batch $b --steps 8 --warmup 2 --cpu-sample 0 --no-secure 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('batch %6d  %8.0f FBS/s  br %7.3f ms (%s) ok=%s' % ($b, d['value'], r['avg_launch_ms'], r['kernel'], d['decrypt_ok']))" || exit 1
  done
done
