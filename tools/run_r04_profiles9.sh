cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | head
python -c "import __graft_entry__ as g; g.smoke()"
timeout 400 python tools/fuzz_parity.py 240 31337 | tail -1
bash tools/profile.sh r04_int8u --low-memory --ld-dtype int8 > gpurun_out/r04_profile9_log.txt 2>&1
bash tools/profile.sh r04_fast_int8u --math fast --low-memory --ld-dtype int8 >> gpurun_out/r04_profile9_log.txt 2>&1
bash tools/profile.sh r04_mix --model mixture >> gpurun_out/r04_profile9_log.txt 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out -name "*.db" -delete; find gpurun_out -name "*agent_info*" -delete
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
python tools/show_bench.py gpurun_out/r04_bench_default.json | cut -c1-200 | head -10
