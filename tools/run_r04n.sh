cd $GRAFT_REPO_ROOT
L="viprs_amd/lib/libviprs_hip.so build/libviprs_hip_tcpl1.so"
for t in 12 16; do for a in "" "fast"; do echo "== TEAM0=$t $a"; VIPRS_TEAM0=$t timeout 900 python tools/multi_ab.py $L -- $a; done; done > gpurun_out/r04n_ab.txt 2>&1
VIPRS_HIP_LIB=build/libviprs_hip_tcpl1.so timeout 600 python -m pytest tests/test_gpu_farfield.py -m gpu -x -q -k "spike_slab and float32" 2>&1 | tail -3 >> gpurun_out/r04n_ab.txt
