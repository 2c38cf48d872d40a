cd $GRAFT_REPO_ROOT
VIPRS_HIP_LIB=build/libviprs_hip_gring64.so timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -x -q -k "grid" 2>&1 | tail -2
timeout 900 python tools/multi_ab.py viprs_amd/lib/libviprs_hip.so build/libviprs_hip_gring64.so -- grid
