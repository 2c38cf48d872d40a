"""One-shot mixture E-step (K = 4) on one banded component: band kernel (components serial per lane) vs VIPRS_BAND=0."""
import sys, time, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tools.banded_bench import banded
from viprs_amd.plan import LDPlan
from viprs_amd.utils import synthetic as syn
from tests.test_oracle_vs_ref import _mixture_inputs
m, w, K = 100000, 250, 4
for upper in (False, True):
    lb, ip, data = banded(m, w, upper)
    ld = syn.SyntheticLD(lb, ip, data, np.array([0, m]), np.zeros(1), upper, 1.0)
    rng = np.random.default_rng(1)
    ss = syn.SyntheticSumstats((rng.standard_normal(m) * 0.003).astype(np.float32), np.full(m, 1e5), np.zeros(m, np.float32), 1e5)
    inp = syn.make_inputs(ss)
    mix, st0 = _mixture_inputs(ld, ss, K)
    plan = LDPlan(lb, ip, data, upper)
    st = {k: v.copy() for k, v in st0.items()}
    def sweep():
        plan.e_step_mixture(inp.std_beta, st["var_gamma"], st["var_mu"], st["eta"], st["q"], st["eta_diff"], mix["log_null_pi"],
                            mix["u_logs"], mix["shvt"], mix["mu_mult"], 1.0)
    sweep()
    t0 = time.perf_counter()
    for _ in range(3): sweep()
    print(f"mixture K={K} banded m={m} w={w} {'upper' if upper else 'sym'} VIPRS_BAND={os.environ.get('VIPRS_BAND','1')}: {(time.perf_counter()-t0)/3*1e3:.1f} ms per one-shot call")
    plan.close()
