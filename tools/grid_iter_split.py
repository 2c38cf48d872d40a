"""Development tool: one iteration of the batched grid fit taken apart (device calls timed with a synchronisation between
them) on cfg3 with 32 models."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from viprs_amd import _lib
from viprs_amd.plan import DeviceState, LDPlan
from viprs_amd.utils import synthetic as syn
ld, ss, inp = syn.make_problem("cfg3", low_memory=False)
G = 32
plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, False)
st = DeviceState(plan, "float32", "grid", G)
st.upload("std_beta", inp.std_beta)
st.set_n_per_snp(ss.n_per_snp)
st.set_snp_weights(np.full(ld.m, 1.0 / ld.m))
pts = syn.grid_points(G, ld.m)
for g in range(G):
    st.reset_column(g, 0.01)
act = np.arange(G, dtype=np.int32)
rows = np.array([[g, np.log(0.01 / 0.99), np.log(5e4), 0.8, 5e4, 1.0] for g in range(G)], dtype=np.float64)
sync = lambda: _lib.check(_lib.lib.viprs_device_synchronize(0))
T = np.zeros(4)
n = 12
for it in range(n + 3):
    sync(); t0 = time.perf_counter()
    st.prep_columns(rows); sync(); t1 = time.perf_counter()
    st.e_step(ld.dq_scale, active_model_idx=act, sync=False); sync(); t2 = time.perf_counter()
    st.sums_columns_begin(act, np.ones(G)); t3 = time.perf_counter()
    v = st.sums_columns_end(); t4 = time.perf_counter()
    if it >= 3:
        T += (t1 - t0, t2 - t1, t3 - t2, t4 - t3)
print("ms per iteration: prep_columns %.3f  e_step %.3f  sums_begin (enqueue) %.3f  sums_end (run + read back) %.3f" % tuple(T / n * 1e3))
