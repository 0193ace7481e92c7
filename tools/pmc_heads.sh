#!/bin/bash
# L2 behaviour of the job list's sharding: TCC hit / miss / request counts and wave waits with RAYLIB_JOB_HEADS=1 (one head, the whole chip sweeps the
# image together) against 8 (one band per XCD).  usage: tools/pmc_heads.sh <workload> <outdir>
WL=${1:-breakfast_300k_1080p_128spp}
OUT=${2:-/root/repo/gpurun_out/pmc_heads_$WL}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for H in 1 8; do
  export RAYLIB_JOB_HEADS=$H
  timeout 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/h$H/tcc -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --workload $WL > $OUT/h${H}_tcc.log 2>&1
  timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/h$H/fetch -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --workload $WL > $OUT/h${H}_fetch.log 2>&1
  timeout 240 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/h$H/sq -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --workload $WL > $OUT/h${H}_sq.log 2>&1
  echo "== heads $H"; python3 /root/repo/tools/pmc_summarize.py $OUT/h$H | awk '/k_trace/{p=1} /k_resolve/{p=0} p'
done
