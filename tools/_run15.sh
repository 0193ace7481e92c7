cd /root/repo
mkdir -p gpurun_out/r3o
STEPS=10 python tools/gpu_variants.py slowmath base fastinl > gpurun_out/r3o/var_cornell.log 2>&1
cat gpurun_out/r3o/var_cornell.log
WL=breakfast_300k_1080p_128spp STEPS=5 python tools/gpu_variants.py slowmath fastinl > gpurun_out/r3o/var_breakfast.log 2>&1
cat gpurun_out/r3o/var_breakfast.log
