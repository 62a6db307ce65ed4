#!/bin/bash
# rocprofv3 evidence for profiles/rNN: kernel-trace stats, then PMC counters in their own passes (no trace domains mixed
# in), for a list of (name, command) sets; summarised per kernel by tools/profile_summary.py.
#   usage (on the GPU box): bash tools/profile_round.sh <tag> [set ...]      -> gpurun_out/prof_<tag>/<set>/
#   sets: p1024 (bench.py headline), cu256 (bench.py --batch 256: one bootstrap per CU), secure (128-bit p = 15, two key bits per
#         step), secure1 (one key bit per step), p31 (config 5: 128-bit set for p = 31, two key bits per step on whole CUs),
#         p31g1 (the same with one key bit per step), p63 (N = 4096, one key bit per step), p4 (N = 1024 128-bit set),
#         securek2 / p4k2 (GLWE dimension k = 2 at N = 1024: k_blind_rotate_pairs_k2 at the 128-bit sets for p = 15 and p = 4),
#         secure256 / p31cu (the p = 15 and p = 31 sets at one bootstrap per CU), lean512 (bench.py --batch 512: two workgroups per CU),
#         p4k3 / p7k3 (GLWE dimension 3 at N = 512: k_blind_rotate_glwe at the default 128-bit sets for p = 4 and p = 7, two full rounds of 768), k3cu256 (one bootstrap per workgroup),
#         k2cu256 / k2cu512 (the k = 2 set for p = 15 on k_blind_rotate_cu_k2: one bootstrap on the twelve waves of a workgroup, one / two rounds)
TAG=${1:-r04}; shift
SETS=${@:-p1024 cu256 secure p31}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TOP=gpurun_out/prof_$TAG
mkdir -p $TOP
PMCS=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_LDS_IDX_ACTIVE")
run_set() {   # $1 = subdir, rest = command
  local NAME=$1
  local OUT=$TOP/$1; shift
  rm -rf $OUT && mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- "$@" > $OUT/under_rocprof.out 2> $OUT/under_rocprof.err
  local i=0
  for c in "${PMCS[@]}"; do
    i=$((i+1))
    rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$i -- "$@" > /dev/null 2> $OUT/pmc_$i.err
  done
  python3 tools/profile_summary.py $OUT > $OUT/summary.json
  cp $OUT/kt/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
  echo "set $NAME done"
}
for S in $SETS; do
  case $S in
    p1024)   run_set p1024 python3 bench.py --steps 10 --cpu-sample 0 --no-secure ;;
    cu256)   run_set cu256 python3 bench.py --batch 256 --steps 10 --cpu-sample 0 --no-secure ;;
    secure)  run_set secure python3 tools/secure_bench.py 1024 5 15 70 ;;
    secure1) run_set secure1 python3 tools/secure_bench.py 1024 5 15 70 1 ;;
    p31)     run_set p31 python3 tools/secure_bench.py 1024 4 31 325 ;;
    p31g1)   run_set p31g1 python3 tools/secure_bench.py 1024 4 31 325 1 ;;
    p63)     run_set p63 python3 tools/secure_bench.py 1024 3 63 100 ;;
    p4)      run_set p4 python3 tools/secure_bench.py 1024 5 4 2 ;;
    securek2) run_set securek2 python3 tools/secure_bench.py 1024 5 15 70 k2 ;;
    p4k2)    run_set p4k2 python3 tools/secure_bench.py 1024 5 4 2 k2 ;;
    p4k3)    run_set p4k3 python3 tools/secure_bench.py 1536 5 4 2 k3 ;;
    p7k3)    run_set p7k3 python3 tools/secure_bench.py 1536 5 7 10 k3 ;;
    k3cu256) run_set k3cu256 python3 tools/secure_bench.py 256 8 4 2 k3 ;;
    secure256) run_set secure256 python3 tools/secure_bench.py 256 8 15 70 ;;
    k2cu256) run_set k2cu256 python3 tools/secure_bench.py 256 8 15 70 k2 ;;
    k2cu512) run_set k2cu512 python3 tools/secure_bench.py 512 6 15 70 k2 ;;
    p31cu)   run_set p31cu python3 tools/secure_bench.py 256 6 31 325 ;;
    lean512) run_set lean512 python3 bench.py --batch 512 --steps 10 --cpu-sample 0 --no-secure ;;
  esac
done
