#!/bin/bash
# every library in gpurun_exp/ (and the in-tree one) through tools/cu_latency.py, same GPU, same call:  bash tools/cu_variants.sh [steps]
for lib in tfhe_fbs_map_amd/libfbsexec.so gpurun_exp/libfbsexec_*.so; do
  [ -f $lib ] || continue
  FBS_LIB=$PWD/$lib timeout -k 10 200 python3 tools/cu_latency.py "$@" 2>/dev/null || echo "$lib FAILED"
done
