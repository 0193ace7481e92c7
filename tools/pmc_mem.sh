#!/bin/bash
# Memory-pipeline counters of the megakernel (TA / TCP / TD), one --pmc pass each group.  usage: tools/pmc_mem.sh <tag> [ENV=VAL...]
TAG=$1; shift
OUT=/root/repo/gpurun_out/pmcm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
run() { name=$1; shift; timeout 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --workload ${WL:-cornell_1080p_64spp} > $OUT/$name.log 2>&1; }
# few counters per block and pass (a request the hardware cannot schedule aborts rocprofv3 and then hangs it: hence the timeouts)
run ta TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run td TD_TD_BUSY_sum TD_TC_STALL_sum
run grbm GRBM_GUI_ACTIVE
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_trace" in row["Kernel_Name"]:
            acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()): print("   %-36s %.6g" % (c, sum(v) / len(v)))
PY
