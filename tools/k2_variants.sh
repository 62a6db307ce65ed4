#!/bin/bash
# every library in gpurun_exp/ (and the in-tree one) through tools/k2_latency.py, same GPU, same call:  bash tools/k2_variants.sh [shape] [steps] [sizes...]
for lib in tfhe_fbs_map_amd/libfbsexec.so gpurun_exp/libfbsexec_*.so; do
  [ -f $lib ] || continue
  FBS_LIB=$PWD/$lib timeout -k 10 120 python3 tools/k2_latency.py "$@" 2>/dev/null || echo "$lib FAILED"
done
