#!/bin/bash
# gpurun_out/prof_<tag>/ (tools/profile_round.sh) -> profiles/<round>/: per-set summaries + kernel-trace stats, and the PMC record
# bench.py reads (stamped with the SHA-256 of the kernel sources as they are NOW: run right after the profile round).
#   usage (here, after the gpurun call merged its outputs): bash tools/collect_profiles.sh r04 r04
TAG=${1:-r04}; ROUND=${2:-r04}
SRC=gpurun_out/prof_$TAG; DST=profiles/$ROUND
mkdir -p $DST
rm -f $DST/pmc_blind_rotate.json
for S in $(ls $SRC); do
  [ -f $SRC/$S/summary.json ] || continue
  cp $SRC/$S/summary.json $DST/summary_$S.json
  [ -f $SRC/$S/kernel_stats.csv ] && cp $SRC/$S/kernel_stats.csv $DST/kernel_stats_$S.csv
done
[ -f $SRC/bench.json ] && cp $SRC/bench.json $DST/bench.json
rec() { [ -f $DST/summary_$1.json ] && python3 tools/make_pmc_record.py $DST/summary_$1.json $2 $3 $DST/pmc_blind_rotate.json $4 profiles/$ROUND/summary_$1.json; }
rec p1024 630 1 1024
rec cu256 630 1 256
rec secure 714 2 1024
rec secure1 710 1 1024
rec p31 766 2 1024
rec p31g1 766 1 1024
rec p63 822 1 1024
rec p4 638 1 1024
rec securek2 734 2 1024
rec p4k2 630 2 1024
rec p4k3 614 2 1536
rec p7k3 682 2 1536
rec k3cu256 614 2 256
rec k2cu256 734 2 256
rec k2cu512 734 2 512
rec secure256 714 2 256
rec p31cu 766 2 256
rec lean512 630 1 512
