#!/bin/bash
# Quick SQ counter pass for A/B runs: tools/pmc_quick.sh <tag> [env assignments...]  (bench cornell, 2 steps)
TAG=$1; shift
OUT=/root/repo/gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/sq1 -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --workload ${WL:-cornell_1080p_64spp} > $OUT/sq1.log 2>&1
timeout 240 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT SQ_WAIT_ANY --output-format csv -d $OUT/sq2 -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --workload ${WL:-cornell_1080p_64spp} > $OUT/sq2.log 2>&1
python3 /root/repo/tools/pmc_summarize.py $OUT 2>&1 | grep -A40 "k_trace" | grep -v "k_resolve" | head -30
