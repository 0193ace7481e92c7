#!/bin/bash
# Copies what tools/profile_round.sh <tag> (and the validation runs) left under gpurun_out/ into profiles/ (tracked).  usage: tools/collect_profiles.sh r04
TAG=${1:-r05}
SRC=gpurun_out/round_$TAG
cp $SRC/bench_default.json profiles/${TAG}_bench_default.json
for WL in cornell_1080p_64spp breakfast_300k_1080p_128spp breakfast_interior_300k_1080p_128spp breakfast_textured_interior_300k_1080p_128spp; do
  [ -d $SRC/stats_$WL ] || continue
  [ -f $SRC/bench_$WL.json ] && cp $SRC/bench_$WL.json profiles/${TAG}_bench_$WL.json
  cp $(ls -t $SRC/stats_$WL/*/*kernel_stats.csv | head -1) profiles/${TAG}_kernel_stats_$WL.csv
  cp $SRC/pmc_$WL.txt profiles/${TAG}_pmc_$WL.txt
done
cp $SRC/valu_calib.txt profiles/${TAG}_valu_calib.txt
cp $SRC/static_mix.json profiles/${TAG}_static_mix.json
cp $SRC/pmc_traffic.json profiles/pmc_traffic.json
cp $SRC/valu_calib.json profiles/valu_calib.json
for f in dynamic_mix.txt diag_wave_steps.txt vmem_width_bench.txt; do [ -f $SRC/$f ] && cp $SRC/$f profiles/${TAG}_$f; done
[ -f $SRC/dynamic_mix_counts.json ] && cp $SRC/dynamic_mix_counts.json profiles/dynamic_mix_counts.json
[ -f $SRC/valu_classes.json ] && cp $SRC/valu_classes.json profiles/valu_classes.json
[ -f $SRC/pmc_mem_interior.txt ] && cp $SRC/pmc_mem_interior.txt profiles/${TAG}_pmc_mem_interior.txt
[ -f $SRC/vmem_gather_bench.txt ] && cp $SRC/vmem_gather_bench.txt profiles/${TAG}_vmem_gather_bench.txt
ls -la profiles | grep ${TAG}
