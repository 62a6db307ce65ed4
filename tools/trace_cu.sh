#!/bin/bash
# Phase trace of the one-bootstrap-per-CU kernel: builds two variants of the library with -DFBS_CU_TRACE
# (s_memtime at every phase boundary of a step, per wave of workgroup 0, summed over the rotation) -- one without the priority
# flips (-DFBS_CU_PRIO=0), one with -- and runs bench.py --batch 64 on each.   usage (GPU box): bash tools/trace_cu.sh > out.txt
# Columns (cycles): 0 loop top | 1 rotated read + digits | 2 cross stages + re-deal write | 3 WAIT barrier 1 | 4 forward transforms |
#   5 products + hand-over write | 6 WAIT barrier 2 | 7 hand-over read + inverse + write | 8 WAIT barrier 3 | 9 join + accumulate | 10 WAIT barrier 4
cd $(dirname $0)/../tfhe_fbs_map_amd/csrc
for V in 0 2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -ffp-contract=off -DFBS_CU_TRACE -DFBS_CU_PRIO=$V -c -o /tmp/cu_trace$V.o fbs_blind_rotate_cu.hip || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -o /tmp/libfbsexec_trace$V.so build/fbs_host.cpp.o build/fbs_plan.cpp.o build/fbs_capi.cpp.o build/fbs_kernels.hip.o build/fbs_blind_rotate.hip.o build/fbs_blind_rotate_k2.hip.o build/fbs_blind_rotate_glwe.hip.o /tmp/cu_trace$V.o build/fbs_mapper_search.hip.o || exit 1
done
cd ../..
for V in 0 2; do
  echo "== FBS_CU_PRIO=$V, P1024, 64 bootstraps (waves 0-3: component 0, waves 4-7: component 1; wave w and w + 4 share a SIMD)"
  FBS_LIB=/tmp/libfbsexec_trace$V.so python3 bench.py --batch 64 --steps 2 --warmup 1 --cpu-sample 0 --no-secure 2>&1 >/dev/null | grep -A8 "^trace" | tail -9
done
# the two-key-bit kernels (no priority flips).  Columns: 0 loop top | 1 digits + cross stages + re-deal write | 2 WAIT re-deal barrier |
#   3 forward transforms | 4 products (key words fetched pair by pair) + hand-over write | 5 WAIT hand-over barrier |
#   6 hand-over read + inverse + write | 7 WAIT re-deal-back barrier | 8 join + accumulate
echo "== k_blind_rotate_cu_pairs<11,2>: 128-bit set for p = 31, 64 bootstraps"
FBS_LIB=/tmp/libfbsexec_trace2.so python3 tools/secure_bench.py 64 1 31 325 2>&1 | grep -A8 "^trace" | tail -9
echo "== k_blind_rotate_cu_pairs<11,1>: 128-bit set for p = 15, 64 bootstraps"
FBS_LIB=/tmp/libfbsexec_trace2.so python3 tools/secure_bench.py 64 1 15 70 2>&1 | grep -A8 "^trace" | tail -9
