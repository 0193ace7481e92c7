#!/bin/bash
# Everything profiles/ holds for a round, in one call on the GPU box (about 4 minutes):
#   bench JSON lines, rocprofv3 --kernel-trace --stats summaries, PMC passes (separate --pmc runs, no trace domains).
# usage: tools/profile_round.sh <round tag, e.g. r01>
TAG=${1:-r02}
OUT=/root/repo/gpurun_out/round_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for WL in cornell_1080p_64spp breakfast_300k_1080p_128spp; do
  python3 /root/repo/bench.py --steps 5 --warmup 1 --workload $WL > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$WL -- python3 /root/repo/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra --workload $WL > $OUT/stats_$WL.log 2>&1
  /root/repo/tools/pmc_profile.sh $WL $OUT/pmc_$WL > $OUT/pmc_$WL.log 2>&1
  python3 /root/repo/tools/pmc_summarize.py $OUT/pmc_$WL > $OUT/pmc_$WL.txt 2>&1
done
# derived figures (HBM bytes, VALU issue at 2 cycles per wave instruction, lane utilisation) -> gpurun_out/round_$TAG/pmc_traffic.json
cp /root/repo/profiles/pmc_traffic.json $OUT/pmc_traffic_before.json 2>/dev/null
ROUND_TAG=$TAG python3 /root/repo/tools/pmc_traffic.py cornell_1080p_64spp=$OUT/pmc_cornell_1080p_64spp breakfast_300k_1080p_128spp=$OUT/pmc_breakfast_300k_1080p_128spp > $OUT/pmc_traffic.log 2>&1
cp /root/repo/profiles/pmc_traffic.json $OUT/pmc_traffic.json
# the bench lines again, now carrying this round's counters
for WL in cornell_1080p_64spp breakfast_300k_1080p_128spp; do
  python3 /root/repo/bench.py --steps 10 --warmup 2 --workload $WL > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
done
find $OUT -name "*kernel_stats.csv" | head
tail -c 600 $OUT/bench_cornell_1080p_64spp.json; echo; tail -c 600 $OUT/bench_breakfast_300k_1080p_128spp.json
