cd /root/repo
mkdir -p gpurun_out/r3j
timeout 1500 python -m pytest tests/test_gpu_multi_rank.py -x -q -m gpu > gpurun_out/r3j/pytest_multi.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3j/pytest_multi.log
tail -n 15 gpurun_out/r3j/pytest_multi.log
for P in 1 0; do
RAYLIB_PIPELINE=$P RAYLIB_GPU_MAP=0,0 python bench.py --gpus 2 --steps 20 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/r3j/bench_lib2_p$P.json 2> gpurun_out/r3j/bench_lib2_p$P.err
python - <<PY
import json
d=json.load(open('gpurun_out/r3j/bench_lib2_p$P.json'))
print('pipeline $P', d['ms_per_step'], d['ms_per_step_spread'], d['multi_gpu'], d['config']['frame_check'])
PY
done
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3j/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3j/pytest.log
tail -n 6 gpurun_out/r3j/pytest.log
