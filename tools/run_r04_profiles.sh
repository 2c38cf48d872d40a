cd $GRAFT_REPO_ROOT
bash tools/profile.sh r04 > gpurun_out/r04_profile_log.txt 2>&1
bash tools/profile.sh r04_upper --low-memory >> gpurun_out/r04_profile_log.txt 2>&1
bash tools/profile.sh r04_int8u --low-memory --ld-dtype int8 >> gpurun_out/r04_profile_log.txt 2>&1
bash tools/profile.sh r04_mix --model mixture >> gpurun_out/r04_profile_log.txt 2>&1
bash tools/profile.sh r04_grid --model grid >> gpurun_out/r04_profile_log.txt 2>&1
bash tools/profile.sh r04_f64 --precision float64 --low-memory --ld-dtype int8 >> gpurun_out/r04_profile_log.txt 2>&1
bash tools/profile.sh r04_fast --math fast >> gpurun_out/r04_profile_log.txt 2>&1
bash tools/profile.sh r04_fast_int8u --math fast --low-memory --ld-dtype int8 >> gpurun_out/r04_profile_log.txt 2>&1
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
# keep the merged output small: only the csv / json summaries
find gpurun_out -name "*.db" -delete; find gpurun_out -name "*agent_info*" -delete
du -sh gpurun_out | tail -1
