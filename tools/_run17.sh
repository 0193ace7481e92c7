cd /root/repo
mkdir -p gpurun_out/r3q
WL=breakfast_300k_1080p_128spp STEPS=5 python tools/gpu_variants.py prev base nodeptr > gpurun_out/r3q/var_breakfast.log 2>&1
cat gpurun_out/r3q/var_breakfast.log
STEPS=10 python tools/gpu_variants.py prev base > gpurun_out/r3q/var_cornell.log 2>&1
cat gpurun_out/r3q/var_cornell.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r3q/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3q/pytest.log
tail -n 4 gpurun_out/r3q/pytest.log
