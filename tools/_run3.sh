cd /root/repo
mkdir -p gpurun_out/r3c
WL=breakfast_300k_1080p_128spp STEPS=5 python tools/gpu_variants.py base:RAYLIB_JOB_HEADS=1,RAYLIB_GUIDED=0 base:RAYLIB_JOB_HEADS=1,RAYLIB_GUIDED=1 base:RAYLIB_JOB_HEADS=8,RAYLIB_GUIDED=0 base:RAYLIB_JOB_HEADS=8,RAYLIB_GUIDED=1 base:RAYLIB_JOB_HEADS=8,RAYLIB_GUIDED=2 strips:RAYLIB_JOB_HEADS=8,RAYLIB_GUIDED=0 strips:RAYLIB_JOB_HEADS=8,RAYLIB_GUIDED=1 > gpurun_out/r3c/heads_breakfast.log 2>&1
STEPS=5 python tools/gpu_variants.py base:RAYLIB_JOB_HEADS=1,RAYLIB_GUIDED=0 base:RAYLIB_JOB_HEADS=8,RAYLIB_GUIDED=0 base:RAYLIB_JOB_HEADS=8,RAYLIB_GUIDED=1 base:RAYLIB_JOB_HEADS=8,RAYLIB_GUIDED=2 strips:RAYLIB_JOB_HEADS=8,RAYLIB_GUIDED=0 > gpurun_out/r3c/heads_cornell.log 2>&1
cat gpurun_out/r3c/heads_breakfast.log gpurun_out/r3c/heads_cornell.log
