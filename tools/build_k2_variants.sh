#!/bin/bash
# Variant builds of csrc/fbs_blind_rotate_k2.hip into gpurun_exp/ (travels to the GPU box, not into the history):
#   bash tools/build_k2_variants.sh name1 "-DFLAG=1 ..." name2 "-D..."      -> gpurun_exp/libfbsexec_<name>.so
# (the other objects are the in-tree build's: run make first)
cd $(dirname $0)/../tfhe_fbs_map_amd/csrc || exit 1
mkdir -p ../../gpurun_exp
while [ $# -ge 2 ]; do
  NAME=$1; FLAGS=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -ffp-contract=off $FLAGS -c -o /tmp/k2_$NAME.o fbs_blind_rotate_k2.hip || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -o ../../gpurun_exp/libfbsexec_$NAME.so build/fbs_host.cpp.o build/fbs_capi.cpp.o build/fbs_kernels.hip.o \
     build/fbs_blind_rotate.hip.o build/fbs_blind_rotate_cu.hip.o /tmp/k2_$NAME.o build/fbs_mapper_search.hip.o || exit 1
  echo "built gpurun_exp/libfbsexec_$NAME.so ($FLAGS)"
done
