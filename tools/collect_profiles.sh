#!/bin/bash
# Copies what tools/profile_round.sh <tag> (and the validation runs) left under gpurun_out/ into profiles/ (tracked).  usage: tools/collect_profiles.sh r04
TAG=${1:-r04}
SRC=gpurun_out/round_$TAG
cp $SRC/bench_default.json profiles/${TAG}_bench_default.json
for WL in cornell_1080p_64spp breakfast_300k_1080p_128spp breakfast_interior_300k_1080p_128spp; do
  [ -f $SRC/bench_$WL.json ] && cp $SRC/bench_$WL.json profiles/${TAG}_bench_$WL.json
  cp $(ls -t $SRC/stats_$WL/*/*kernel_stats.csv | head -1) profiles/${TAG}_kernel_stats_$WL.csv
  cp $SRC/pmc_$WL.txt profiles/${TAG}_pmc_$WL.txt
done
cp $SRC/valu_calib.txt profiles/${TAG}_valu_calib.txt
cp $SRC/static_mix.json profiles/${TAG}_static_mix.json
cp $SRC/pmc_traffic.json profiles/pmc_traffic.json
cp $SRC/valu_calib.json profiles/valu_calib.json
[ -f $SRC/pmc_mem_interior.txt ] && cp $SRC/pmc_mem_interior.txt profiles/${TAG}_pmc_mem_interior.txt
[ -f $SRC/vmem_gather_bench.txt ] && cp $SRC/vmem_gather_bench.txt profiles/${TAG}_vmem_gather_bench.txt
ls -la profiles | grep ${TAG}
