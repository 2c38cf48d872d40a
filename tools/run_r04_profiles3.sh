cd $GRAFT_REPO_ROOT
bash tools/profile.sh r04_grid --model grid > gpurun_out/r04_profile3_log.txt 2>&1
bash tools/profile.sh r04_grid_int8 --model grid --ld-dtype int8 >> gpurun_out/r04_profile3_log.txt 2>&1
bash tools/profile.sh r04_grid_upper --model grid --low-memory >> gpurun_out/r04_profile3_log.txt 2>&1
bash tools/profile.sh r04_grid_int8u --model grid --low-memory --ld-dtype int8 >> gpurun_out/r04_profile3_log.txt 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out -name "*.db" -delete; find gpurun_out -name "*agent_info*" -delete
python tools/grid_iter_split.py 2>&1 | tail -2
