cd /root/repo
mkdir -p gpurun_out/r3r
WL=breakfast_300k_1080p_128spp STEPS=5 python tools/gpu_variants.py base pf > gpurun_out/r3r/var_breakfast.log 2>&1
cat gpurun_out/r3r/var_breakfast.log
RAYLIB_LIB=/root/repo/software-raytracing_amd/libraylib_pf.so timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r3r/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3r/pytest.log
tail -n 3 gpurun_out/r3r/pytest.log
python tools/gpu_scenes_time.py all 2>&1 | grep triangles > gpurun_out/r3r/scenes_base.log
RAYLIB_LIB=/root/repo/software-raytracing_amd/libraylib_pf.so python tools/gpu_scenes_time.py all 2>&1 | grep triangles > gpurun_out/r3r/scenes_pf.log
cat gpurun_out/r3r/scenes_base.log gpurun_out/r3r/scenes_pf.log
