cd /root/repo
mkdir -p gpurun_out/r3h
STEPS=10 python tools/gpu_variants.py base boxv2 rootrcp0 > gpurun_out/r3h/var_cornell.log 2>&1
cat gpurun_out/r3h/var_cornell.log
RAYLIB_LIB=/root/repo/software-raytracing_amd/libraylib_stamps.so python tools/gpu_stamps.py 2>&1 | grep -v "^Initialize\|rank 0 of" > gpurun_out/r3h/stamps1.log
RAYLIB_LIB=/root/repo/software-raytracing_amd/libraylib_stamps2.so python tools/gpu_stamps.py 2>&1 | grep -v "^Initialize\|rank 0 of" > gpurun_out/r3h/stamps2.log
cat gpurun_out/r3h/stamps1.log gpurun_out/r3h/stamps2.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r3h/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3h/pytest.log
RAYLIB_LIB=/root/repo/software-raytracing_amd/libraylib_boxv2.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r3h/pytest_boxv2.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3h/pytest_boxv2.log
tail -4 gpurun_out/r3h/pytest.log gpurun_out/r3h/pytest_boxv2.log
