cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python tools/multi_ab.py viprs_amd/lib/libviprs_hip.so build/libviprs_hip_wide.so --
timeout 600 python tools/multi_ab.py viprs_amd/lib/libviprs_hip.so build/libviprs_hip_wide.so -- upper
timeout 600 python tools/multi_ab.py viprs_amd/lib/libviprs_hip.so build/libviprs_hip_wide.so -- int8 upper
