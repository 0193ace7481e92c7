cd /root/repo
mkdir -p gpurun_out/r3s
STEPS=10 python tools/gpu_variants.py base camlds > gpurun_out/r3s/var_cornell.log 2>&1
cat gpurun_out/r3s/var_cornell.log
tools/verify_fastmath > gpurun_out/r3s/verify_fastmath.txt 2>&1
