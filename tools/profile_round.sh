#!/bin/bash
# rocprofv3 evidence for profiles/rNN: kernel-trace stats, then PMC counters in their own passes (no trace domains mixed
# in), for the headline batch (bench.py, P1024) and for the 128-bit parameter set (tools/secure_bench.py); summarised per
# kernel by tools/profile_summary.py.
#   usage (on the GPU box): bash tools/profile_round.sh [tag]      -> gpurun_out/prof_<tag>/{p1024,secure}/
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TOP=gpurun_out/prof_$TAG
rm -rf $TOP && mkdir -p $TOP
PMCS=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES")
run_set() {   # $1 = subdir, rest = command
  local OUT=$TOP/$1; shift
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- "$@" > $OUT/under_rocprof.out 2> $OUT/under_rocprof.err
  local i=0
  for c in "${PMCS[@]}"; do
    i=$((i+1))
    rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$i -- "$@" > /dev/null 2> $OUT/pmc_$i.err
  done
  python3 tools/profile_summary.py $OUT > $OUT/summary.json
}
run_set p1024 python3 bench.py --steps 10 --cpu-sample 0 --no-secure
run_set secure python3 tools/secure_bench.py 1024 5
python3 bench.py > $TOP/bench.json 2> $TOP/bench.err
tail -c 400 $TOP/bench.json
