cd $GRAFT_REPO_ROOT
timeout 900 python tools/multi_ab.py viprs_amd/lib/libviprs_hip.so build/libviprs_hip_gmask.so -- grid
timeout 900 python tools/multi_ab.py viprs_amd/lib/libviprs_hip.so build/libviprs_hip_gmask.so -- grid int8
