cd /root/repo
mkdir -p gpurun_out/r3a
(cd /tmp && export TMPDIR=/tmp && rocprofv3 -L > /root/repo/gpurun_out/r3a/counters.txt 2>&1)
./tools/valu_calib gpurun_out/r3a/valu_calib.json > gpurun_out/r3a/valu_calib.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r3a/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3a/pytest.log
STEPS=5 python tools/gpu_variants.py base:RAYLIB_JOB_HEADS=1 base:RAYLIB_JOB_HEADS=8 base:RAYLIB_JOB_HEADS=4 > gpurun_out/r3a/heads_cornell.log 2>&1
WL=breakfast_300k_1080p_128spp STEPS=5 python tools/gpu_variants.py base:RAYLIB_JOB_HEADS=1 base:RAYLIB_JOB_HEADS=8 base:RAYLIB_JOB_HEADS=4 > gpurun_out/r3a/heads_breakfast.log 2>&1
tail -5 gpurun_out/r3a/pytest.log; cat gpurun_out/r3a/heads_cornell.log gpurun_out/r3a/heads_breakfast.log; tail -45 gpurun_out/r3a/valu_calib.log
