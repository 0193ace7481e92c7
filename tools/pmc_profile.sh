#!/bin/bash
# PMC passes for the dominant kernel (k_trace).  Separate passes: FETCH_SIZE takes 3 of the 4 TCC
# slots and WRITE_SIZE 2 (MI355X_MICROARCH.md, rocprofv3 PMC slots).  No tracing domains are combined
# with --pmc, and every pass runs under `timeout`: asked for FETCH_SIZE and WRITE_SIZE in ONE pass rocprofv3 aborts ("exceeds the capabilities of the
# hardware") and then sits in its signal handler until it is killed -- 50 GPU-minutes of round 3 went that way.  Usage: tools/pmc_profile.sh <workload> <outdir>
WL=${1:-cornell_1080p_64spp}
OUT=${2:-/root/repo/gpurun_out/pmc_$WL}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --workload $WL > $OUT/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
# the vector memory path's request rate (round 5): one TCP access per lane and load instruction -- the walk's other ceiling (tools/vmem_width_bench.hip: 1.14 per clock and CU)
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR
run grbm GRBM_GUI_ACTIVE
# VALU instructions by class: tools/pmc_traffic.py weights them with the issue costs of tools/valu_calib.hip
run cls1 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
run cls2 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
find $OUT -name "*counter_collection.csv" | head
