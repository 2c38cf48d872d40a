"""float64 state: in-kernel timeline of workgroup 0 (the largest block) -- needs a library built with
-DVIPRS_TILE_PROFILE (make -C viprs_amd/csrc OBJDIR=/tmp/tprof_obj OUT=$PWD/build/libviprs_tprof.so
EXTRA_CXXFLAGS=-DVIPRS_TILE_PROFILE) and VIPRS_HIP_LIB pointing at it.
    VIPRS_HIP_LIB=build/libviprs_tprof.so python tools/fp64_timeline.py [int8|float32] [upper|sym] [block size]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from viprs_amd.plan import DeviceState, LDPlan
from viprs_amd.utils import synthetic as syn

dt = np.dtype(sys.argv[1] if len(sys.argv) > 1 else "int8")
upper = len(sys.argv) > 2 and sys.argv[2] == "upper"
b = int(sys.argv[3]) if len(sys.argv) > 3 else 1536
ld, ss, inp = syn.make_problem(sizes=[b], low_memory=upper, ld_dtype=dt, seed=3, float_precision=np.float64)
plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, upper)
ds = DeviceState(plan, "float64", "spike_slab", 1)
for k in ("std_beta", "u_logs", "sqrt_half_var_tau", "mu_mult"):
    ds.upload(k, getattr(inp, k))
ds.reset(inp.pi)
ds.e_step(ld.dq_scale)
ds.close(); plan.close()
