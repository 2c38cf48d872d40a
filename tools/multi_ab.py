"""A/B/C... of several builds of the library on the same box, alternating, kernel ms p50 of 30 sweeps each:
    python tools/multi_ab.py LIB_A LIB_B [LIB_C ...] -- [upper] [int8|int16] [f64] [grid|mix [K=n]] [uniform] [fast] [sizes=a,b,..] [shard=N]
(each library runs in its own subprocess, 3 rounds; LIB may carry environment switches: path,VIPRS_TEAM0=16)"""
import os, subprocess, sys
sep = sys.argv.index('--') if '--' in sys.argv else len(sys.argv)
libs = sys.argv[1:sep]
args = sys.argv[sep + 1:]
code = r'''
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from viprs_amd.plan import DeviceState, LDPlan
from viprs_amd.utils import synthetic as syn
upper = "upper" in sys.argv; dt = np.int8 if "int8" in sys.argv else np.int16 if "int16" in sys.argv else np.float32
sizes = np.full(1700, 650) if "uniform" in sys.argv else None      # uniform: 1700 blocks of 650 SNPs (no large-block tail)
for a in sys.argv:
    if a.startswith("sizes="): sizes = np.array([int(x) for x in a[6:].split(",")])      # sizes=6000: one isolated block
    if a.startswith("shard="):                                                           # shard=8: rank 0's share of cfg3 sharded 8 ways
        from viprs_amd.parallel import shard_blocks
        all_sizes = syn.block_sizes("cfg3"); sizes = all_sizes[shard_blocks(all_sizes, int(a[6:])) == 0]
ld, ss, inp = syn.make_problem("cfg3", low_memory=upper, ld_dtype=dt, sizes=sizes)
model = "grid" if "grid" in sys.argv else "mixture" if "mix" in sys.argv else "spike_slab"
width = {"grid": 32, "mixture": 4, "spike_slab": 1}[model]
for a in sys.argv:
    if a.startswith("K="): width = int(a[2:])          # mix K=8: another number of components
plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, upper, math_mode="fast" if "fast" in sys.argv else "exact"); ft = "float64" if "f64" in sys.argv else "float32"     # f64: float64 state (the tile kernels)
ds = DeviceState(plan, ft, model, width)
_up = ds.upload; ds.upload = lambda k, a: _up(k, np.asarray(a, dtype=ft, order="F" if (model == "grid" and np.ndim(a) == 2) else "C"))
ds.upload("std_beta", inp.std_beta)
active, pi0 = None, inp.pi
if model == "spike_slab":
    for k in ("u_logs", "sqrt_half_var_tau", "mu_mult"): ds.upload(k, getattr(inp, k))
else:
    extra = syn.make_mixture_inputs(ss, width) if model == "mixture" else syn.make_grid_inputs(ss, width)
    pi0 = extra.pop("pi")
    for k, a in extra.items(): ds.upload(k, a)
    if model == "grid": active = np.arange(width, dtype=np.int32)
pi0 = np.asarray(pi0, dtype=ft) if np.ndim(pi0) else pi0
for _ in range(5): ds.reset(pi0); ds.e_step(ld.dq_scale, active, sync=False)
ds.synchronize(); plan.timing_reset()
import time
t0 = time.perf_counter()
for _ in range(30): ds.reset(pi0); ds.e_step(ld.dq_scale, active, sync=False)
ds.synchronize(); wall = (time.perf_counter() - t0) / 30 * 1e3
t = np.array(plan.timing_history(which=0 if ft == 'float64' else 1))      # (float64: whole-sweep bracket)
print("%.4f %.4f %.4f  wall ms/step %.4f" % (np.median(t), np.percentile(t, 10), np.percentile(t, 90), wall))
'''
for rnd in range(3):
    for name, lib in zip("ABCDEFGH", libs):
        path, *switches = lib.split(",")
        env = dict(os.environ, VIPRS_HIP_LIB=os.path.abspath(path), **dict(x.split("=", 1) for x in switches))
        out = subprocess.run([sys.executable, "-c", code] + args, env=env, capture_output=True, text=True)
        print(rnd, name, os.path.basename(lib), out.stdout.strip() or out.stderr[-300:], flush=True)
