#!/bin/bash
# every library in gpurun_exp/ (and the in-tree one) through tools/glwe_time.py, same GPU, same call:  bash tools/glwe_variants.sh [steps] [batches...]
for lib in tfhe_fbs_map_amd/libfbsexec.so gpurun_exp/libfbsexec_*.so; do
  [ -f $lib ] || continue
  FBS_LIB=$PWD/$lib timeout -k 10 300 python3 tools/glwe_time.py "$@" 2>/dev/null || echo "$lib FAILED"
done
