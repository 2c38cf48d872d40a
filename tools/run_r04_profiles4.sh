cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_float64.py tests/test_gpu_edge_cases.py tests/test_gpu_models.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED"
bash tools/profile.sh r04_f64 --precision float64 --low-memory --ld-dtype int8 > gpurun_out/r04_profile4_log.txt 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out -name "*.db" -delete; find gpurun_out -name "*agent_info*" -delete
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
python tools/show_bench.py gpurun_out/r04_bench_default.json | cut -c1-200 | head -12
python tools/fp64_bench.py cfg3 float32 upper 2>&1 | grep float64
python tools/fp64_bench.py cfg3 int8 upper 4 2>&1 | grep float64
python tools/fp64_bench.py cfg3 int8 sym 2>&1 | grep float64
