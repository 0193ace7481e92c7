cd /root/repo
mkdir -p gpurun_out/r3final
timeout 1500 python -m pytest tests -x -q -m gpu -s > gpurun_out/r3final/pytest_gpu_s.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3final/pytest_gpu_s.log
tail -n 4 gpurun_out/r3final/pytest_gpu_s.log
FUZZ_TMIN=1 timeout 900 python tools/gpu_fuzz.py 2000 20261004 > gpurun_out/r3final/fuzz_2000x11.log 2>&1; tail -n 2 gpurun_out/r3final/fuzz_2000x11.log
bash tools/profile_round.sh r03 > gpurun_out/round_r03.log 2>&1
tail -c 600 gpurun_out/round_r03/bench_default.json; echo
python tools/gpu_scenes_time.py all 2>&1 | grep triangles > gpurun_out/r3final/scenes.log; cat gpurun_out/r3final/scenes.log
