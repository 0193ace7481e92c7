cd /root/repo
mkdir -p gpurun_out/r3i
WL=breakfast_300k_1080p_128spp STEPS=5 python tools/gpu_variants.py base stackt base:RAYLIB_POOL_SHORT_STACK=0 stackt:RAYLIB_POOL_SHORT_STACK=0 > gpurun_out/r3i/var_breakfast.log 2>&1
cat gpurun_out/r3i/var_breakfast.log
RAYLIB_LIB=/root/repo/software-raytracing_amd/libraylib_stackt.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r3i/pytest_stackt.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3i/pytest_stackt.log
tail -n 4 gpurun_out/r3i/pytest_stackt.log
python tools/gpu_scenes_time.py all > gpurun_out/r3i/scenes_base.log 2>&1
RAYLIB_LIB=/root/repo/software-raytracing_amd/libraylib_stackt.so python tools/gpu_scenes_time.py all > gpurun_out/r3i/scenes_stackt.log 2>&1
grep -h "triangles" gpurun_out/r3i/scenes_base.log gpurun_out/r3i/scenes_stackt.log
