cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_farfield.py tests/test_gpu_fullsize_parity.py tests/test_gpu_edge_cases.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python tools/mixed_blocks_bench.py 2>&1 | tail -7
timeout 600 python tools/mixed_blocks_bench.py upper 2>&1 | tail -7
timeout 600 python tools/multi_ab.py viprs_amd/lib/libviprs_hip.so build/libviprs_hip_wide.so --
timeout 600 python tools/multi_ab.py viprs_amd/lib/libviprs_hip.so build/libviprs_hip_wide.so -- int8 upper
