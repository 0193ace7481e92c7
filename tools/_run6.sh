cd /root/repo
mkdir -p gpurun_out/r3f
STEPS=10 python tools/gpu_variants.py boxv1 base noslp > gpurun_out/r3f/var_cornell.log 2>&1
cat gpurun_out/r3f/var_cornell.log
WL=breakfast_300k_1080p_128spp STEPS=5 python tools/gpu_variants.py base noslp > gpurun_out/r3f/var_breakfast.log 2>&1
cat gpurun_out/r3f/var_breakfast.log
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3f/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3f/pytest.log
tail -15 gpurun_out/r3f/pytest.log
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d /root/repo/gpurun_out/r3f/calib_pmc -- /root/repo/tools/valu_calib > /root/repo/gpurun_out/r3f/calib_pmc.log 2>&1)
