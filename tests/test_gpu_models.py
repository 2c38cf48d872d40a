"""GPU: mixture and grid E-steps (panel kernels with the model policies) against the oracle."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H
from tests.test_oracle_vs_ref import _grid_inputs, _mixture_inputs
from viprs_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu
STATE = ("var_gamma", "var_mu", "eta", "q", "eta_diff")


def _run_mix(mod, ld, inp, mix, st0, sweeps, **kw):
    st = {k: v.copy() for k, v in st0.items()}
    for _ in range(sweeps):
        mod.cpp_e_step_mixture(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, inp.std_beta, st["var_gamma"], st["var_mu"],
                               st["eta"], st["q"], st["eta_diff"], mix["log_null_pi"], mix["u_logs"], mix["shvt"],
                               mix["mu_mult"], ld.dq_scale, 1, ld.low_memory, **kw)
    return st


@pytest.mark.parametrize("low_memory", [False, True])
@pytest.mark.parametrize("K", [1, 4, 8, 10])          # K <= 8: panel kernels; K = 10: generic kernel
def test_mixture_matches_oracle(gpu, K, low_memory):
    from viprs_amd.vi import e_step_hip as S
    ld, ss, inp = syn.make_problem(sizes=[70, 1400, 333], low_memory=low_memory, seed=31)
    mix, st0 = _mixture_inputs(ld, ss, K)
    ref = _run_mix(O, ld, inp, mix, st0, 2)
    got = _run_mix(S, ld, inp, mix, st0, 2)
    for k in STATE:
        H.assert_close(got[k], ref[k], 1e-5, k)
    if not low_memory:
        H.assert_state_equal(got, ref)


@pytest.mark.parametrize("low_memory", [False, True])
def test_grid_matches_oracle(gpu, low_memory):
    from viprs_amd.vi import e_step_hip as S
    ld, ss, inp = syn.make_problem(sizes=[130, 1300, 64], low_memory=low_memory, seed=33)
    g, st0 = _grid_inputs(ld, ss, 32)
    active = np.array([31, 0, 7, 8, 21], dtype=np.int32)[::1]
    out = {}
    for name, mod in (("ref", O), ("hip", S)):
        st = {k: v.copy(order="F") for k, v in st0.items()}
        for _ in range(2):
            mod.cpp_e_step_grid(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, inp.std_beta, st["var_gamma"], st["var_mu"],
                                st["eta"], st["q"], st["eta_diff"], g["u_logs"], g["hvt"], g["mu_mult"], ld.dq_scale,
                                active, 1, low_memory)
        out[name] = st
    for k in STATE:
        H.assert_close(out["hip"][k], out["ref"][k], 1e-5, k)
    if not low_memory:
        H.assert_state_equal(out["hip"], out["ref"])
    untouched = [c for c in range(32) if c not in active]
    assert np.all(out["hip"]["eta"][:, untouched] == 0)
