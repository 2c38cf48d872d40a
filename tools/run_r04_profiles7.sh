cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -k "mix or Mix or fit" 2>&1 | grep -E "passed|failed|FAILED"
timeout 600 python tools/fuzz_parity.py 180 9000 | tail -1
python tools/chain_ns_models.py 3619 2>&1 | grep mixture
bash tools/profile.sh r04_mix --model mixture > gpurun_out/r04_profile7_log.txt 2>&1
bash tools/profile.sh r04_mix10u --model mixture --width 10 --low-memory >> gpurun_out/r04_profile7_log.txt 2>&1
bash tools/profile.sh r04_mix20u --model mixture --width 20 --low-memory >> gpurun_out/r04_profile7_log.txt 2>&1
bash tools/profile.sh r04_mix10 --model mixture --width 10 >> gpurun_out/r04_profile7_log.txt 2>&1
bash tools/profile.sh r04_mix20 --model mixture --width 20 >> gpurun_out/r04_profile7_log.txt 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out -name "*.db" -delete; find gpurun_out -name "*agent_info*" -delete
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
python tools/show_bench.py gpurun_out/r04_bench_default.json | cut -c1-200 | head -4
