cd $GRAFT_REPO_ROOT
bash tools/profile.sh r04_mix10u --model mixture --width 10 --low-memory > gpurun_out/r04_profile2_log.txt 2>&1
bash tools/profile.sh r04_mix20u --model mixture --width 20 --low-memory >> gpurun_out/r04_profile2_log.txt 2>&1
bash tools/profile.sh r04_mix10 --model mixture --width 10 >> gpurun_out/r04_profile2_log.txt 2>&1
bash tools/profile.sh r04_mix20 --model mixture --width 20 >> gpurun_out/r04_profile2_log.txt 2>&1
bash tools/profile.sh r04_grid_int8 --model grid --ld-dtype int8 >> gpurun_out/r04_profile2_log.txt 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out -name "*.db" -delete; find gpurun_out -name "*agent_info*" -delete
