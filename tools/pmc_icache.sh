#!/bin/bash
# Instruction-cache counters of the megakernel.  usage: tools/pmc_icache.sh <tag> [ENV=VAL...]
TAG=$1; shift
OUT=/root/repo/gpurun_out/pmci_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
timeout 150 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d $OUT/ic -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --workload ${WL:-cornell_1080p_64spp} > $OUT/ic.log 2>&1
python3 /root/repo/tools/pmc_summarize.py $OUT | grep -A5 "k_trace"
