"""Per-workgroup block timeline of one sweep (needs a -DVIPRS_SWEEP_TRACE build of panel_f32):
    make -C viprs_amd/csrc OBJDIR=../../build/obj_trace OUT=../../build/libviprs_hip_trace.so EXTRA_CXXFLAGS=-DVIPRS_SWEEP_TRACE
    VIPRS_HIP_LIB=build/libviprs_hip_trace.so python tools/sweep_trace.py [upper]
Prints, per 50 us slice of the sweep: busy workgroups, LD bytes being worked on (block bytes spread evenly over the
block's residence), and the tail."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from viprs_amd import _lib as L
from viprs_amd.plan import DeviceState, LDPlan
from viprs_amd.utils import synthetic as syn
upper = "upper" in sys.argv
dt = np.int8 if "int8" in sys.argv else np.float32
ld, ss, inp = syn.make_problem("cfg3", low_memory=upper, ld_dtype=dt)
plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, upper)
ds = DeviceState(plan)
ds.upload("std_beta", inp.std_beta)
for k in ("u_logs", "sqrt_half_var_tau", "mu_mult"):
    ds.upload(k, getattr(inp, k))
buf = (ctypes.c_ulonglong * (4 << 15))()
fn = getattr(L.lib, "viprs_debug_sweep_trace_" + ("i8" if dt == np.int8 else "f32"))
for it in range(4):
    ds.reset(0.01); ds.e_step(ld.dq_scale)
    n = fn(buf, 1 << 15)
a = np.frombuffer(buf, dtype=np.uint64)[: 4 * n].reshape(n, 4).astype(np.int64)
wg, team, b, t0, t1 = a[:, 0] >> 32, a[:, 0] & 1, a[:, 1], a[:, 2] * 10.0, a[:, 3] * 10.0       # ns
T0 = t0.min(); t0 -= T0; t1 -= T0
print(f"records {n}, sweep {t1.max() / 1e3:.1f} us, kernel ms {plan.last_kernel_ms(1):.3f}")
print(f"team records {int(team.sum())}: end of team work at {t1[team == 1].max() / 1e3:.1f} us; largest block {b.max()} "
      f"resident {((t1 - t0)[b == b.max()]).max() / 1e3:.1f} us")
es = np.dtype(dt).itemsize
for lo in np.arange(0, t1.max(), 50e3):
    hi = lo + 50e3
    ov = np.clip(np.minimum(t1, hi) - np.maximum(t0, lo), 0, None)
    busy = ov.sum() / 50e3
    # a team block's record appears once per member: count its bytes once (members share the stream)
    w = np.where(team == 1, 1.0 / np.maximum(1, np.array([np.sum((b == bb) & (team == 1)) / max(1, np.sum((ld.block_start[1:] - ld.block_start[:-1]) == bb)) for bb in b])), 1.0)
    gb = (ov / np.maximum(t1 - t0, 1) * (b.astype(float) ** 2) * es * (0.5 if upper else 1.0) * w).sum() / 1e9
    print(f"  {lo / 1e3:6.0f}-{hi / 1e3:6.0f} us: busy workgroups {busy:6.1f}   stream {gb / 50e-6 / 1e3:6.2f} TB/s   "
          f"blocks in flight: median size {np.median(b[ov > 0]) if (ov > 0).any() else 0:.0f}")
