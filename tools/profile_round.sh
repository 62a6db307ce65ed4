#!/bin/bash
# rocprofv3 evidence for profiles/rNN: kernel-trace stats of the default bench.py run, then PMC counters in their own
# passes (no trace domains mixed in), summarised per kernel by tools/profile_summary.py.
#   usage (on the GPU box): bash tools/profile_round.sh [tag]      -> gpurun_out/prof_<tag>/
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 10 --cpu-sample 0 > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
  d=$OUT/pmc_$(echo $c | cut -c1-14 | tr " " _)
  rocprofv3 --pmc $c --output-format csv -d $d -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2> $d.err
done
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 tools/profile_summary.py $OUT > $OUT/summary.json
cat $OUT/summary.json
