cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_farfield.py tests/test_gpu_fullsize_parity.py tests/test_grid.py -m gpu -x -q -k "grid" 2>&1 | tail -3
timeout 900 python tools/multi_ab.py build/libviprs_hip_epiold.so viprs_amd/lib/libviprs_hip.so -- grid upper
timeout 900 python tools/multi_ab.py build/libviprs_hip_epiold.so viprs_amd/lib/libviprs_hip.so -- grid upper int8
