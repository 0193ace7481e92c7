#!/bin/bash
# Where waves wait / what they issue: SQ activity counters of the megakernel.  usage: tools/pmc_wait.sh <tag> [ENV=VAL...]
TAG=$1; shift
OUT=/root/repo/gpurun_out/pmcw_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
timeout 150 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT --output-format csv -d $OUT/a -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --workload ${WL:-cornell_1080p_64spp} > $OUT/a.log 2>&1
timeout 150 rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/b -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --workload ${WL:-cornell_1080p_64spp} > $OUT/b.log 2>&1
python3 /root/repo/tools/pmc_summarize.py $OUT | awk '/k_trace/{p=1} /k_resolve/{p=0} p'
