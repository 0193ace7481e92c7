#!/bin/bash
# Copies what tools/profile_round.sh <tag> (and the validation runs) left under gpurun_out/ into profiles/ (tracked).  usage: tools/collect_profiles.sh r03
TAG=${1:-r03}
SRC=gpurun_out/round_$TAG
cp $SRC/bench_default.json profiles/${TAG}_bench_default.json
cp $SRC/bench_breakfast_300k_1080p_128spp.json profiles/${TAG}_bench_breakfast_300k_1080p_128spp.json
for WL in cornell_1080p_64spp breakfast_300k_1080p_128spp; do
  cp $(ls -t $SRC/stats_$WL/*/*kernel_stats.csv | head -1) profiles/${TAG}_kernel_stats_$WL.csv
  cp $SRC/pmc_$WL.txt profiles/${TAG}_pmc_$WL.txt
done
cp $SRC/valu_calib.txt profiles/${TAG}_valu_calib.txt
cp $SRC/static_mix.json profiles/${TAG}_static_mix.json
cp $SRC/pmc_traffic.json profiles/pmc_traffic.json
cp $SRC/valu_calib.json profiles/valu_calib.json
[ -f gpurun_out/r3final/pytest_gpu_s.log ] && cp gpurun_out/r3final/pytest_gpu_s.log profiles/${TAG}_pytest_gpu.log
[ -f gpurun_out/r3final/fuzz_2000x12.log ] && cp gpurun_out/r3final/fuzz_2000x12.log profiles/${TAG}_fuzz_2000x12.log
[ -f gpurun_out/r3final/scenes.log ] && cp gpurun_out/r3final/scenes.log profiles/${TAG}_scenes.log
ls -la profiles | grep ${TAG}
[ -f gpurun_out/r3final/verify_fastdiv.txt ] && cp gpurun_out/r3final/verify_fastdiv.txt profiles/${TAG}_verify_fastdiv.txt
[ -f gpurun_out/r3final/verify_fastmath.txt ] && cp gpurun_out/r3final/verify_fastmath.txt profiles/${TAG}_verify_fastmath.txt
