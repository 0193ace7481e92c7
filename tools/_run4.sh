cd /root/repo
mkdir -p gpurun_out/r3d
RAYLIB_LIB=/root/repo/software-raytracing_amd/libraylib_tl.so python tools/gpu_timeline_pool.py > gpurun_out/r3d/timeline_pool.log 2>&1
cat gpurun_out/r3d/timeline_pool.log | grep -v "^Initialize\|rank 0 of"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d /root/repo/gpurun_out/r3d/calib_pmc -- /root/repo/tools/valu_calib > /root/repo/gpurun_out/r3d/calib_pmc.log 2>&1)
tools/pmc_profile.sh cornell_1080p_64spp /root/repo/gpurun_out/r3d/pmc_cornell > gpurun_out/r3d/pmc_cornell.log 2>&1
tools/pmc_profile.sh breakfast_300k_1080p_128spp /root/repo/gpurun_out/r3d/pmc_breakfast > gpurun_out/r3d/pmc_breakfast.log 2>&1
python3 tools/pmc_summarize.py gpurun_out/r3d/pmc_cornell > gpurun_out/r3d/pmc_cornell.txt 2>&1
python3 tools/pmc_summarize.py gpurun_out/r3d/pmc_breakfast > gpurun_out/r3d/pmc_breakfast.txt 2>&1
