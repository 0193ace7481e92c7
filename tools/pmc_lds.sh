#!/bin/bash
# LDS counter pass on the bench workload: tools/pmc_lds.sh <tag> [env assignments...]
TAG=$1; shift
OUT=/root/repo/gpurun_out/pmcl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
timeout 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_LDS_UNALIGNED_STALL --output-format csv -d $OUT/l1 -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --workload ${WL:-cornell_1080p_64spp} > $OUT/l1.log 2>&1
timeout 240 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --output-format csv -d $OUT/l2 -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --workload ${WL:-cornell_1080p_64spp} > $OUT/l2.log 2>&1
python3 /root/repo/tools/pmc_summarize.py $OUT 2>&1 | grep -A40 "k_trace" | grep -v "k_resolve" | head -30
tail -3 $OUT/l1.log | cut -c1-300
