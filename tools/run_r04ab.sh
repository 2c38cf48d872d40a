cd $GRAFT_REPO_ROOT
timeout 900 python tools/multi_ab.py viprs_amd/lib/libviprs_hip.so build/libviprs_hip_gepi3.so build/libviprs_hip_gepi2.so -- grid upper
VIPRS_HIP_LIB=build/libviprs_hip_gepi3.so timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_farfield.py -m gpu -x -q -k "grid" 2>&1 | tail -2
