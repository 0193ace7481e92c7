#!/bin/bash
# Which hardware counter tracks the VALU issue port's busy time?  The single-opcode streams of tools/valu_calib (known cost per opcode: profiles/valu_calib.json)
# under the SQ counters that might: per opcode, counter / SQ_INSTS_VALU against the calibrated cycles per instruction (tools/valu_busy_pmc.py).
# Every pass on its own and under timeout (a counter the hardware cannot schedule aborts rocprofv3 and then hangs it).  usage: tools/valu_busy_pmc.sh <outdir>
OUT=${1:-/root/repo/gpurun_out/valu_busy}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 60 rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u | grep -E "VALU|BUSY|CYCLES" > $OUT/sq_counters_available.txt
run() { name=$1; shift; timeout 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- /root/repo/tools/valu_calib > $OUT/$name.log 2>&1 || echo "pass $name failed or timed out" >> $OUT/failed.txt; }
run a GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
run b SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run c SQ_INSTS_VALU SQ_INST_CYCLES_VALU
run d SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU
python3 /root/repo/tools/valu_busy_pmc.py $OUT /root/repo/profiles/valu_calib.json
