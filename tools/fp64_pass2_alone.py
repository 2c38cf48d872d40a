"""float64 state, upper-triangular LD: a workload of equal-size blocks of one class (no second stream, the second pass
runs alone on the whole chip) -- run under rocprofv3 --kernel-trace --stats to read the second pass's own rate.
    python tools/fp64_pass2_alone.py [blocks] [size] [int8|float32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from viprs_amd.plan import DeviceState, LDPlan
from viprs_amd.utils import synthetic as syn

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dt = np.dtype(sys.argv[3] if len(sys.argv) > 3 else "int8")
ld, ss, inp = syn.make_problem(sizes=[size] * nb, low_memory=True, ld_dtype=dt, kind="ar1", float_precision=np.float64)
plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, True)
ds = DeviceState(plan, "float64", "spike_slab", 1)
for k in ("std_beta", "u_logs", "sqrt_half_var_tau", "mu_mult"):
    ds.upload(k, getattr(inp, k))
for _ in range(12):
    ds.reset(inp.pi); ds.e_step(ld.dq_scale, sync=False)
ds.synchronize()
print(f"{nb} x {size} {dt.name}: upper triangle {nb * size * (size - 1) / 2 * dt.itemsize / 1e6:.0f} MB")
