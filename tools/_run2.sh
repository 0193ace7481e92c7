cd /root/repo
mkdir -p gpurun_out/r3b
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d /root/repo/gpurun_out/r3b/calib_pmc -- /root/repo/tools/valu_calib > /root/repo/gpurun_out/r3b/calib_pmc.log 2>&1)
tools/pmc_heads.sh breakfast_300k_1080p_128spp /root/repo/gpurun_out/r3b/heads > gpurun_out/r3b/heads.txt 2>&1
cat gpurun_out/r3b/heads.txt
