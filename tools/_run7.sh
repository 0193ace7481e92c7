cd /root/repo
mkdir -p gpurun_out/r3g
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 200 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit cycles --pc-sampling-method stochastic --pc-sampling-interval 1048576 --output-format csv -d /root/repo/gpurun_out/r3g/pcs -- python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /root/repo/gpurun_out/r3g/pcs.log 2>&1
echo "stochastic exit $?"
ls -la /root/repo/gpurun_out/r3g/pcs/* | head
tail -5 /root/repo/gpurun_out/r3g/pcs.log
if ! ls /root/repo/gpurun_out/r3g/pcs/*/*pc_sampling* >/dev/null 2>&1; then
timeout 200 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval 100 --output-format csv -d /root/repo/gpurun_out/r3g/pch -- python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /root/repo/gpurun_out/r3g/pch.log 2>&1
echo "host_trap exit $?"
ls -la /root/repo/gpurun_out/r3g/pch/* | head
tail -5 /root/repo/gpurun_out/r3g/pch.log
fi
cd /root/repo/gpurun_out/r3g && for f in */*/*pc_sampl*.csv; do echo $f; head -3 $f; wc -l $f; done 2>/dev/null | head -30
# keep the upload small
find /root/repo/gpurun_out/r3g -name "*.csv" -size +40M -delete
