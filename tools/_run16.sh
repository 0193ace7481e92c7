cd /root/repo
bash tools/profile_round.sh r03 > gpurun_out/round_r03.log 2>&1
tail -n 30 gpurun_out/round_r03.log
