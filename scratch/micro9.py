import sys, time, numpy as np
sys.path.insert(0, '.')
from viprs_amd.plan import LDPlan, DeviceState
from viprs_amd.utils import synthetic as syn
from tests.test_oracle_vs_ref import _grid_inputs, _mixture_inputs
for cfg in ("cfg2", "cfg3"):
    ld, ss, inp = syn.make_problem(cfg, low_memory=False)
    plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, False)
    mix, st0 = _mixture_inputs(ld, ss, 4)
    st = DeviceState(plan, "float32", "mixture", 4)
    st.upload("std_beta", inp.std_beta); st.upload("log_null_pi", mix["log_null_pi"]); st.upload("u_logs", mix["u_logs"]); st.upload("sqrt_half_var_tau", mix["shvt"]); st.upload("mu_mult", mix["mu_mult"])
    for k in ("var_gamma","var_mu","eta","q","eta_diff"): st.upload(k, st0[k])
    for _ in range(2): st.e_step(ld.dq_scale)
    plan.timing_reset()
    for _ in range(5): st.e_step(ld.dq_scale)
    ms = np.median(plan.timing_history(0)); print(cfg, "mixture K=4:", ms, "ms", ld.m/ms/1e3, "M SNP-updates/s")
    st.close()
    if cfg == "cfg2":
        G = 32
        g, gs = _grid_inputs(ld, ss, G)
        sg = DeviceState(plan, "float32", "grid", G)
        sg.upload("std_beta", inp.std_beta); sg.upload("u_logs", g["u_logs"]); sg.upload("half_var_tau", g["hvt"]); sg.upload("mu_mult", g["mu_mult"])
        for k in ("var_gamma","var_mu","eta","q","eta_diff"): sg.upload(k, gs[k])
        for _ in range(2): sg.e_step(ld.dq_scale)
        plan.timing_reset()
        for _ in range(3): sg.e_step(ld.dq_scale)
        ms = np.median(plan.timing_history(0)); print(cfg, "grid G=32:", ms, "ms", ld.m*G/ms/1e3, "M SNP-model-updates/s")
        sg.close()
    plan.close()
