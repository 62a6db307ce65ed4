#!/bin/bash
# Variant builds of one kernel source into gpurun_exp/ (travels to the GPU box, not into the history):
#   bash tools/build_variants.sh fbs_blind_rotate_k2.hip name1 "-DFLAG=1 ..." name2 "-D..."      -> gpurun_exp/libfbsexec_<name>.so
# (the other objects are the in-tree build's: run make first)
cd $(dirname $0)/../tfhe_fbs_map_amd/csrc || exit 1
mkdir -p ../../gpurun_exp
SRC=$1; shift
OBJS=""
for f in fbs_host.cpp fbs_plan.cpp fbs_capi.cpp fbs_kernels.hip fbs_blind_rotate.hip fbs_blind_rotate_cu.hip fbs_blind_rotate_k2.hip fbs_blind_rotate_glwe.hip fbs_mapper_search.hip; do
  [ $f = $SRC ] || OBJS="$OBJS build/$f.o"
done
while [ $# -ge 2 ]; do
  NAME=$1; FLAGS=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -ffp-contract=off $FLAGS -c -o /tmp/variant_$NAME.o $SRC || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -o ../../gpurun_exp/libfbsexec_$NAME.so $OBJS /tmp/variant_$NAME.o || exit 1
  echo "built gpurun_exp/libfbsexec_$NAME.so ($SRC $FLAGS)"
done
