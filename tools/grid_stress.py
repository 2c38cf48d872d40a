"""Stress: N batched grid sweeps on cfg3 (32 models) from the same start must be bit-identical -- the team blocks' a-vector
granules of one launch meet the next launch (generation tags), two sweeps without a reset in between per iteration.
    python tools/grid_stress.py N [upper] [int8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from viprs_amd.plan import DeviceState, LDPlan
from viprs_amd.utils import synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
upper = "upper" in sys.argv
dt = np.int8 if "int8" in sys.argv else np.float32
ld, ss, inp = syn.make_problem("cfg3", low_memory=upper, ld_dtype=dt, kind="longrange")
G = 32
plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, upper)
st = DeviceState(plan, "float32", "grid", G)
extra = syn.make_grid_inputs(ss, G)
pi0 = extra.pop("pi")
st.upload("std_beta", inp.std_beta)
for k, a in extra.items():
    st.upload(k, a)
act = np.arange(G, dtype=np.int32)
first, bad = None, 0
for it in range(n):
    st.reset(pi0)
    st.e_step(ld.dq_scale, act)
    st.e_step(ld.dq_scale, act)
    cur = {k: st.download(k) for k in ("var_gamma", "eta", "q", "eta_diff")}
    if first is None:
        first = cur
    elif any(not np.array_equal(cur[k], first[k]) for k in cur):
        bad += 1
        k = next(k for k in cur if not np.array_equal(cur[k], first[k]))
        rows = np.nonzero((cur[k] != first[k]).any(axis=1))[0]
        blk = np.searchsorted(ld.block_start, rows[:5], side="right") - 1
        print(f"sweep {it}: {k} differs in {rows.size} SNPs, first blocks {sorted(set(int(x) for x in blk))} sizes {[int(ld.block_start[x+1]-ld.block_start[x]) for x in sorted(set(blk))]}", flush=True)
print(f"{bad} of {n - 1} iterations differ from the first")
