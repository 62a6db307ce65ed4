#!/bin/bash
# Phase trace of the k = 2 latency shape k_blind_rotate_cu_k2: a variant library with -DFBS_CU_TRACE (cycle stamps at
# every phase boundary of a step, per wave of workgroup 0, summed over the rotation), run at the shipped set on 64 and 256 bootstraps.
#   usage (GPU box, after tools/build_variants.sh fbs_blind_rotate_k2.hip trace "-DFBS_CU_TRACE"): bash tools/trace_k2.sh
# Columns (cycles): 0 loop top | 1 psi look-ups issued, digits, cross stages, re-deal write | 2 WAIT barrier 1 | 3 read + forward transform
#   (second register pair's key words requested inside) | 4 monomial factors, products, hand-over write | 5 WAIT barrier 2 |
#   6 hand-over read, next step's key request, inverse transform, write | 7 WAIT barrier 3 | 8 read, joining stages, accumulate
for B in 64 256; do
  echo "== $B bootstraps"
  FBS_LIB=$PWD/gpurun_exp/libfbsexec_trace.so python3 tools/secure_bench.py $B 1 15 70 k2 2>&1 | grep -A12 "^trace" | tail -13
done
