cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -k "grid or Grid" 2>&1 | grep -E "passed|failed|FAILED"
timeout 300 python tools/grid_stress.py 60 | tail -1
bash tools/profile.sh r04_grid --model grid > gpurun_out/r04_profile8_log.txt 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out -name "*.db" -delete; find gpurun_out -name "*agent_info*" -delete
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
python tools/show_bench.py gpurun_out/r04_bench_default.json | cut -c1-200 | head -4
