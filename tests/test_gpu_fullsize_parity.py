"""GPU: the WHOLE 1.1 M-SNP state of BASELINE configs[2..4] against the oracle, bit for bit, in BOTH LD forms.

One sweep from the standard start (var_gamma = pi, everything else 0) over the genome-wide workload
(1 700 LD blocks, 953 M LD entries of the generator's long-range, non-Toeplitz kind: every entry of every
block changes the result, see tests/test_gpu_farfield.py): spike-and-slab, the K = 4 sparse mixture and the 32-model grid
(oracle on 4 of its columns -- the models of a grid are independent).  The symmetric form AND the
upper-triangular form, which is the reference's default (`low_memory=True`, VIPRS.py:75).  The
single-threaded oracle needs a few seconds per case on the GPU box's host.
"""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H
from viprs_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[False, True], ids=["symmetric", "upper"])
def genome(request, gpu):
    from viprs_amd.plan import LDPlan
    low_memory = request.param
    ld, ss, inp = syn.make_problem("cfg3", low_memory=low_memory, kind="longrange")
    plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, low_memory)
    yield ld, ss, inp, plan
    plan.close()


def _sweep(plan, model, width, uploads, pi0, dq, active=None):
    from viprs_amd.plan import DeviceState
    state = DeviceState(plan, "float32", model, width)
    for name, arr in uploads.items():
        state.upload(name, arr)
    state.reset(pi0)
    state.e_step(dq, active)
    got = {k: state.download(k) for k in H.STATE}
    state.close()
    return got


def test_cfg3_spike_slab_whole_state_equals_oracle(genome):
    ld, ss, inp, plan = genome
    assert ld.m > 1_000_000 and len(ld.block_start) - 1 == 1700
    ref = H.run_oracle(ld, inp, inp.state_copy(), sweeps=1)
    got = _sweep(plan, "spike_slab", 1, {k: getattr(inp, k) for k in ("std_beta", "u_logs", "sqrt_half_var_tau", "mu_mult")},
                 inp.pi, ld.dq_scale)
    H.assert_state_equal(got, ref)
    assert plan.last_skipped() == int((ref["eta_diff"] == 0).sum())


def test_cfg3_mixture_k4_whole_state_equals_oracle(genome):
    ld, ss, inp, plan = genome
    K = 4
    x = syn.make_mixture_inputs(ss, K)
    pi0 = x.pop("pi")
    got = _sweep(plan, "mixture", K, dict(std_beta=inp.std_beta, **x), pi0, ld.dq_scale)
    vg = np.full((ld.m, K), pi0, dtype=np.float32)
    vm = np.zeros((ld.m, K), dtype=np.float32)
    eta, q, ed = (np.zeros(ld.m, dtype=np.float32) for _ in range(3))
    O.cpp_e_step_mixture(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, inp.std_beta, vg, vm, eta, q, ed,
                         x["log_null_pi"], x["u_logs"], x["sqrt_half_var_tau"], x["mu_mult"], ld.dq_scale, 1,
                         ld.low_memory)
    H.assert_state_equal(got, dict(var_gamma=vg, var_mu=vm, eta=eta, q=q, eta_diff=ed))


def test_cfg3_grid_32_models_equal_oracle_on_all_columns(genome):
    ld, ss, inp, plan = genome
    G = 32
    x = syn.make_grid_inputs(ss, G)
    pi0 = x.pop("pi")
    got = _sweep(plan, "grid", G, dict(std_beta=inp.std_beta, **x), pi0, ld.dq_scale, np.arange(G, dtype=np.int32))
    cols = np.arange(G, dtype=np.int32)                         # all 32 models of the 4 x 8 grid (~30 s of oracle)
    mk = lambda: np.asfortranarray(np.zeros((ld.m, G), dtype=np.float32))
    vg = np.asfortranarray(np.full((ld.m, G), pi0, dtype=np.float32))
    vm, eta, q, ed = mk(), mk(), mk(), mk()
    O.cpp_e_step_grid(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, inp.std_beta, vg, vm, eta, q, ed, x["u_logs"],
                      x["half_var_tau"], x["mu_mult"], ld.dq_scale, cols, 1, ld.low_memory)
    ref = dict(var_gamma=vg, var_mu=vm, eta=eta, q=q, eta_diff=ed)
    for k in H.STATE:
        for g in cols:
            assert np.array_equal(got[k][:, g], ref[k][:, g]), (k, int(g), int((got[k][:, g] != ref[k][:, g]).sum()))


def test_cfg3_int8_upper_whole_state_equals_oracle(gpu):
    """The combination the reference runs by default on its published LD stores: int8-quantised LD
    (dq_scale = 1 / 127), upper-triangular form (second pass on coalesced 32-byte row pieces)."""
    from viprs_amd.plan import LDPlan
    ld, ss, inp = syn.make_problem("cfg3", low_memory=True, ld_dtype=np.int8, kind="longrange")
    plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, True)
    try:
        ref = H.run_oracle(ld, inp, inp.state_copy(), sweeps=1)
        got = _sweep(plan, "spike_slab", 1, {k: getattr(inp, k) for k in ("std_beta", "u_logs", "sqrt_half_var_tau", "mu_mult")},
                     inp.pi, ld.dq_scale)
        H.assert_state_equal(got, ref)
    finally:
        plan.close()
