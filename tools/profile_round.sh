#!/bin/bash
# Everything profiles/ holds for a round, in one call on the GPU box (about 4 minutes):
#   bench JSON lines, rocprofv3 --kernel-trace --stats summaries, PMC passes (separate --pmc runs, no trace domains).
# usage: tools/profile_round.sh <round tag, e.g. r01>
TAG=${1:-r01}
OUT=/root/repo/gpurun_out/round_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for WL in cornell_1080p_64spp breakfast_300k_1080p_128spp; do
  python3 /root/repo/bench.py --steps 5 --warmup 1 --workload $WL > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$WL -- python3 /root/repo/bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload $WL > $OUT/stats_$WL.log 2>&1
  /root/repo/tools/pmc_profile.sh $WL $OUT/pmc_$WL > $OUT/pmc_$WL.log 2>&1
  python3 /root/repo/tools/pmc_summarize.py $OUT/pmc_$WL > $OUT/pmc_$WL.txt 2>&1
done
find $OUT -name "*kernel_stats.csv" | head
tail -c 600 $OUT/bench_cornell_1080p_64spp.json; echo; tail -c 600 $OUT/bench_breakfast_300k_1080p_128spp.json
