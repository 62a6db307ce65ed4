#!/bin/bash
# A/B of kernel variants on the N = 2048 shapes: every library in gpurun_exp/ against the product build, same GPU
for lib in tfhe_fbs_map_amd/libfbsexec.so gpurun_exp/*.so; do
  echo "== $lib"
  FBS_LIB=$PWD/$lib timeout -k 10 120 python3 tools/secure_bench.py 1024 5
  FBS_LIB=$PWD/$lib timeout -k 10 120 python3 tools/config5_bench.py
done
