cd $GRAFT_REPO_ROOT
L="build/libviprs_hip_base4.so build/libviprs_hip_nowait.so"
for a in "" "int8 upper" "fast" "fast int8 upper"; do echo "== $a"; timeout 900 python tools/multi_ab.py $L -- $a; done > gpurun_out/r04m_ab.txt 2>&1
