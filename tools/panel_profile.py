"""In-kernel phase timeline of the team panel kernel (block 0 of the largest class, members 0 and 1).
Needs a library built with -DVIPRS_PANEL_PROFILE:
    make -C viprs_amd/csrc OBJDIR=../../build/obj_prof OUT=../../build/libviprs_hip_prof.so EXTRA_CXXFLAGS=-DVIPRS_PANEL_PROFILE
    VIPRS_HIP_LIB=build/libviprs_hip_prof.so python tools/panel_profile.py [cfg3 | SIZE xN]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from viprs_amd.plan import DeviceState, LDPlan
from viprs_amd.utils import synthetic as syn

UPPER = "upper" in sys.argv
argv = [a for a in sys.argv if a != "upper"]
arg = argv[1] if len(argv) > 1 else "cfg3"
if arg == "cfg3":
    ld, ss, inp = syn.make_problem("cfg3", low_memory=UPPER)
else:
    ld, ss, inp = syn.make_problem(sizes=[int(arg)] * int(sys.argv[2] if len(sys.argv) > 2 else 1), low_memory=False, seed=3)
plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, ld.low_memory)
ds = DeviceState(plan)
ds.upload("std_beta", inp.std_beta)
for k in ("u_logs", "sqrt_half_var_tau", "mu_mult"):
    ds.upload(k, getattr(inp, k))
for it in range(2):
    print(f"---- sweep {it}", flush=True)
    ds.reset(0.01)
    ds.e_step(1.0)
print("kernel ms", plan.last_kernel_ms(1))
