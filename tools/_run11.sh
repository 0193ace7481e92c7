cd /root/repo
mkdir -p gpurun_out/r3k
WL=breakfast_300k_1080p_128spp STEPS=5 python tools/gpu_variants.py base wgchunk:RAYLIB_JOB_CHUNK=256 wgchunk:RAYLIB_JOB_CHUNK=512 wgchunk:RAYLIB_JOB_CHUNK=1024 > gpurun_out/r3k/var_breakfast.log 2>&1
cat gpurun_out/r3k/var_breakfast.log
for P in 1 0; do
RAYLIB_PIPELINE=$P RAYLIB_GPU_MAP=0,0 python bench.py --gpus 2 --steps 40 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/r3k/bench_lib2_p$P.json 2> gpurun_out/r3k/bench_lib2_p$P.err
RAYLIB_PIPELINE=$P RAYLIB_GPU_MAP=0,0,0,0,0,0,0,0 python bench.py --gpus 8 --steps 40 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/r3k/bench_lib8_p$P.json 2> gpurun_out/r3k/bench_lib8_p$P.err
python - <<PY
import json
for n in (2,8):
    d=json.load(open('gpurun_out/r3k/bench_lib%d_p$P.json'%n))
    print('pipeline $P ranks',n, 'ms/step %.3f'%d['ms_per_step'], 'gather %.3f scatter %.3f'%(d['multi_gpu']['gather_ms'], d['multi_gpu']['scatter_ms']), [round(x,2) for x in d['multi_gpu']['rank_kernel_ms']], d['config']['frame_check'])
PY
done
RAYLIB_LIB=/root/repo/software-raytracing_amd/libraylib_stamps.so python tools/gpu_stamps.py breakfast 2>&1 | grep -v "^Initialize\|rank 0 of" > gpurun_out/r3k/stamps_breakfast.log; tail -n 12 gpurun_out/r3k/stamps_breakfast.log
