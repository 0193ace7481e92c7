#!/bin/bash
# Everything profiles/ holds for a round, in one call on the GPU box (about 6 minutes):
#   VALU issue costs (tools/valu_calib under timeout 240 rocprofv3 --pmc), bench JSON lines, timeout 240 rocprofv3 --kernel-trace --stats summaries,
#   PMC passes (separate --pmc runs, no trace domains), the derived figures (tools/pmc_traffic.py) and the bench lines that replay them.
# usage: tools/profile_round.sh <round tag, e.g. r03>      (then copy gpurun_out/round_<tag>/... into profiles/, see profiles/README.md)
TAG=${1:-r05}
OUT=/root/repo/gpurun_out/round_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# issue cycles per wave64 VALU instruction, by opcode, in counted cycles
timeout 240 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/calib_pmc -- /root/repo/tools/valu_calib > $OUT/valu_calib_timing.log 2>&1
python3 /root/repo/tools/valu_calib_pmc.py $OUT/calib_pmc /root/repo/profiles/valu_calib.json > $OUT/valu_calib.txt 2>&1
cp /root/repo/profiles/valu_calib.json $OUT/valu_calib.json
# which hardware class counter counts which opcode (what tools/dynamic_mix.py classifies with)
bash /root/repo/tools/valu_class_pmc.sh $OUT/valu_classes > /dev/null 2>&1
python3 /root/repo/tools/valu_class_pmc.py $OUT/valu_classes > $OUT/valu_classes.json 2>/dev/null && cp $OUT/valu_classes.json /root/repo/profiles/valu_classes.json
for WL in cornell_1080p_64spp breakfast_300k_1080p_128spp breakfast_interior_300k_1080p_128spp breakfast_textured_interior_300k_1080p_128spp; do
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$WL -- python3 /root/repo/bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-extra --workload $WL > $OUT/stats_$WL.log 2>&1
  /root/repo/tools/pmc_profile.sh $WL $OUT/pmc_$WL > $OUT/pmc_$WL.log 2>&1
  python3 /root/repo/tools/pmc_summarize.py $OUT/pmc_$WL > $OUT/pmc_$WL.txt 2>&1
done
# derived figures (HBM bytes, class-weighted VALU issue, lane utilisation), stamped with the library's build id -> gpurun_out/round_$TAG/pmc_traffic.json
cp /root/repo/profiles/pmc_traffic.json $OUT/pmc_traffic_before.json 2>/dev/null
python3 /root/repo/tools/static_mix.py > $OUT/static_mix.json 2>$OUT/static_mix.err
ROUND_TAG=$TAG python3 /root/repo/tools/pmc_traffic.py cornell_1080p_64spp=$OUT/pmc_cornell_1080p_64spp breakfast_300k_1080p_128spp=$OUT/pmc_breakfast_300k_1080p_128spp breakfast_interior_300k_1080p_128spp=$OUT/pmc_breakfast_interior_300k_1080p_128spp breakfast_textured_interior_300k_1080p_128spp=$OUT/pmc_breakfast_textured_interior_300k_1080p_128spp > $OUT/pmc_traffic.log 2>&1
# wave-level step counts of the level-2 diagnostic build (libraylib_stamps2.so: make variant VARIANT=stamps2 EXTRA=-DRL_DIAG_STAMPS=2) and the mix weighted by them
if [ -f /root/repo/software-raytracing_amd/libraylib_stamps2.so ]; then
  for w in cornell breakfast interior textured; do RAYLIB_LIB=/root/repo/software-raytracing_amd/libraylib_stamps2.so python3 /root/repo/tools/gpu_stamps.py $w 2>&1 | grep -E "diagnostic slots|trace " | sed "s/^/$w: /"; done > $OUT/diag_wave_steps.txt 2>&1
  python3 /root/repo/tools/dynamic_mix_counts.py $OUT/diag_wave_steps.txt > $OUT/dynamic_mix_counts.log 2>&1
  cp /root/repo/profiles/dynamic_mix_counts.json $OUT/dynamic_mix_counts.json
fi
python3 /root/repo/tools/dynamic_mix.py cornell_1080p_64spp,breakfast_300k_1080p_128spp,breakfast_interior_300k_1080p_128spp,breakfast_textured_interior_300k_1080p_128spp > $OUT/dynamic_mix.txt 2>&1
cp /root/repo/profiles/pmc_traffic.json $OUT/pmc_traffic.json
# the bench lines, now carrying this round's counters: the default invocation (headline + extra + cpu baseline) and the second workload on its own
python3 /root/repo/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 /root/repo/bench.py --steps 20 --warmup 2 --workload breakfast_300k_1080p_128spp > $OUT/bench_breakfast_300k_1080p_128spp.json 2> $OUT/bench_breakfast_300k_1080p_128spp.err
python3 /root/repo/bench.py --steps 10 --warmup 2 --workload breakfast_interior_300k_1080p_128spp > $OUT/bench_breakfast_interior_300k_1080p_128spp.json 2> $OUT/bench_breakfast_interior_300k_1080p_128spp.err
python3 /root/repo/bench.py --steps 10 --warmup 2 --workload breakfast_textured_interior_300k_1080p_128spp > $OUT/bench_breakfast_textured_interior_300k_1080p_128spp.json 2> $OUT/bench_breakfast_textured_interior_300k_1080p_128spp.err
timeout 120 /root/repo/tools/vmem_width_bench > $OUT/vmem_width_bench.txt 2>&1
# the memory pipeline's counters of the interior frame (TA / TCP / TD) and the gather microbenchmark they are read against
WL=breakfast_interior_300k_1080p_128spp bash /root/repo/tools/pmc_mem.sh round_${TAG}_interior > $OUT/pmc_mem_interior.txt 2>&1
timeout 120 /root/repo/tools/vmem_gather_bench > $OUT/vmem_gather_bench.txt 2>&1
find $OUT -name "*kernel_stats.csv" | head
tail -c 1500 $OUT/bench_default.json; echo
