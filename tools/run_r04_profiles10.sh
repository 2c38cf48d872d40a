cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | head
timeout 500 python tools/fuzz_parity.py 240 8086 | tail -1
timeout 300 python tools/stress_repro.py 150 --mixture 2>&1 | tail -1
bash tools/profile.sh r04_mix10u --model mixture --width 10 --low-memory > gpurun_out/r04_profile10_log.txt 2>&1
bash tools/profile.sh r04_mix20u --model mixture --width 20 --low-memory >> gpurun_out/r04_profile10_log.txt 2>&1
bash tools/profile.sh r04_mix10 --model mixture --width 10 >> gpurun_out/r04_profile10_log.txt 2>&1
bash tools/profile.sh r04_mix20 --model mixture --width 20 >> gpurun_out/r04_profile10_log.txt 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out -name "*.db" -delete; find gpurun_out -name "*agent_info*" -delete
