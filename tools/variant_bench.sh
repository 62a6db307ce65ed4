#!/bin/bash
# run bench.py against every experimental library in gpurun_exp/ (same GPU) + a parity spot check
for lib in tfhe_fbs_map_amd/libfbsexec.so gpurun_exp/*.so; do
  FBS_LIB=$PWD/$lib LIBNAME=$lib timeout -k 10 200 python tools/variant_one.py
done
