import sys, time, numpy as np
sys.path.insert(0, '.')
from viprs_amd.plan import LDPlan, DeviceState
from viprs_amd.utils import synthetic as syn

def run(sizes, math="exact", low_memory=False, reps=5, ld_dtype=np.float32):
    ld = syn.make_ld(np.array(sizes), low_memory=low_memory, ld_dtype=ld_dtype, seed=5)
    ss = syn.make_sumstats(ld, seed=5); inp = syn.make_inputs(ss)
    plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, ld.low_memory, math_mode=math)
    st = DeviceState(plan)
    for n in ("std_beta","u_logs","sqrt_half_var_tau","mu_mult"): st.upload(n, getattr(inp,n))
    for _ in range(2): st.reset(inp.pi); st.e_step(ld.dq_scale)
    plan.timing_reset()
    for _ in range(reps): st.reset(inp.pi); st.e_step(ld.dq_scale)
    ms = np.median(plan.timing_history(1))
    m = ld.m; nnz = int(ld.ld_indptr[-1]) * (2 if low_memory else 1)
    print(f"{math:5s} lm={int(low_memory)} {ld_dtype.__name__:7s} nblk={len(sizes):5d} b={sizes[0]:5d} m={m:8d}: {ms*1e3:9.1f} us  {ms*1e6/m:8.1f} ns/SNP  {nnz*np.dtype(ld_dtype).itemsize/ms/1e6:9.1f} GB/s")
    plan.close()

for math in ("exact", "fast"):
    for b in (64, 128, 256, 512, 1024, 2048, 3619):
        run([b], math)
for math in ("exact", "fast"):
    for nb, b in ((256, 650), (768, 650), (1536, 650), (3072, 650), (768, 256), (256, 2048)):
        run([b]*nb, math)
run([650]*1536, "exact", low_memory=True)
run([650]*1536, "exact", ld_dtype=np.int8)
