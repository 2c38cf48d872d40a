cd $GRAFT_REPO_ROOT
bash tools/profile.sh r04_grid_upper --model grid --low-memory > gpurun_out/r04_profile5_log.txt 2>&1
bash tools/profile.sh r04_grid_int8u --model grid --low-memory --ld-dtype int8 >> gpurun_out/r04_profile5_log.txt 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out -name "*.db" -delete; find gpurun_out -name "*agent_info*" -delete
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
python tools/show_bench.py gpurun_out/r04_bench_default.json | cut -c1-200 | head -10
