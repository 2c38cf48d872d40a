# A/B of two builds on one box (round 5): base = build/libviprs_hip_base.so, new = the in-tree library
cd $GRAFT_REPO_ROOT
A=build/libviprs_hip_base.so; B=viprs_amd/lib/libviprs_hip.so
for args in "$@"; do echo "== $args"; python tools/multi_ab.py $A $B -- $args; done
