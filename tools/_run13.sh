cd /root/repo
mkdir -p gpurun_out/r3m
timeout 600 python -m pytest tests/test_math_exact.py -x -q -m gpu -s > gpurun_out/r3m/pytest_math.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3m/pytest_math.log
tail -n 8 gpurun_out/r3m/pytest_math.log
STEPS=10 python tools/gpu_variants.py prev base > gpurun_out/r3m/var_cornell.log 2>&1
cat gpurun_out/r3m/var_cornell.log
WL=breakfast_300k_1080p_128spp STEPS=5 python tools/gpu_variants.py prev base > gpurun_out/r3m/var_breakfast.log 2>&1
cat gpurun_out/r3m/var_breakfast.log
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3m/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3m/pytest.log
tail -n 5 gpurun_out/r3m/pytest.log
