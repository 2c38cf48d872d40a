"""GPU: parity on LD whose FAR field matters.

Analytic AR(1) blocks (rho <= 0.8) cannot tell a correct far-from-diagonal update from a missing one:
entries more than ~2 panels off the diagonal are below half an ulp of `q` (zeroing every |i - j| > 128
entry leaves all five state vectors bit-identical).  The inputs here are the generator's non-Toeplitz
long-range blocks ("longrange": every entry O(0.05-0.3)) and SURVEY 8d's "realistic" variant ("sample":
sample correlation of n = 4 b simulated genotypes, noise ~ 1/sqrt(n) everywhere) -- every test first
PROVES, with the oracle, that cutting the far field changes the result, so it can never go blind again.

Covered: single-workgroup blocks (700, 1 400), team blocks (1 700, 2 400, 3 619, 6 000, 13 000), both LD
forms, fp32 / int8 / int16 LD, spike-and-slab, mixture K in {4, 10, 20}, grid (batched matrix-core
kernel and the per-(block, model) item schedule).  All comparisons are `==` on the five state vectors
(e_step.hpp:157-175, 307-338, 387-440: the upper form's `dot` is the sequential fma chain in the
reference build, which oracle/_ref confirms bit for bit on this data: tests/test_oracle_vs_ref.py).
"""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H
from tests.test_oracle_vs_ref import _grid_inputs, _mixture_inputs
from viprs_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu


def assert_far_field_matters(ld, inp, width=128):
    """Oracle self-check: zeroing every LD entry more than `width` columns off the diagonal CHANGES the result."""
    ref = H.run_oracle(ld, inp, inp.state_copy())
    cut = H.run_oracle(H.cut_far_field(ld, width), inp, inp.state_copy())
    changed = int((cut["q"] != ref["q"]).sum())
    assert changed > ld.m // 2, f"far field does not matter for this input: only {changed}/{ld.m} q entries change"


SS_CASES = [
    # (sizes, kind, ld_dtype, sweeps)
    ([700, 1400, 90], "longrange", np.float32, 2),           # single workgroups
    ([700, 1400, 90], "sample", np.float32, 2),
    ([1700, 2400, 65], "longrange", np.float32, 2),          # teams of 4 / 8
    ([1700, 650], "sample", np.float32, 1),
    ([2400], "sample", np.float32, 1),
    ([3619, 650, 1536], "longrange", np.float32, 2),         # cfg3's largest block
    ([6000, 77], "longrange", np.float32, 1),                # BASELINE's clip limit
    ([13000], "longrange", np.float32, 1),
    ([700, 1400, 1700], "longrange", np.int8, 2),
    ([2400, 3619], "longrange", np.int8, 1),
    ([1400, 2400], "longrange", np.int16, 2),
    ([1700, 650], "sample", np.int8, 1),
    # populous team classes: more large blocks than teams fit (team size drops 12 -> 8 -> 4, several blocks per team),
    # and the team budget split between a populous large and a populous medium class (launch_panel.inc)
    ([2400] * 30 + [300] * 40, "longrange", np.int8, 1),
    ([2500] * 45 + [2000] * 60 + [500] * 30, "longrange", np.int8, 1),
]


@pytest.mark.parametrize("low_memory", [False, True], ids=["symmetric", "upper"])
@pytest.mark.parametrize("sizes, kind, ld_dtype, sweeps", SS_CASES,
                         ids=[f"{'-'.join(map(str, c[0][:3]))}{'-x' + str(len(c[0])) if len(c[0]) > 3 else ''}_{c[1]}_{np.dtype(c[2]).name}"
                              for c in SS_CASES])
def test_spike_slab_far_field(gpu, sizes, kind, ld_dtype, sweeps, low_memory):
    ld, ss, inp = syn.make_problem(sizes=sizes, low_memory=low_memory, ld_dtype=ld_dtype, seed=61, kind=kind)
    assert_far_field_matters(ld, inp)
    st0 = inp.state_copy()
    H.assert_state_equal(H.run_hip(ld, inp, st0, sweeps=sweeps), H.run_oracle(ld, inp, st0, sweeps=sweeps))


def _run_mix(mod, ld, inp, mix, st0, sweeps):
    st = {k: v.copy() for k, v in st0.items()}
    for _ in range(sweeps):
        mod.cpp_e_step_mixture(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, inp.std_beta, st["var_gamma"], st["var_mu"],
                               st["eta"], st["q"], st["eta_diff"], mix["log_null_pi"], mix["u_logs"], mix["shvt"],
                               mix["mu_mult"], ld.dq_scale, 1, ld.low_memory)
    return st


@pytest.mark.parametrize("low_memory", [False, True], ids=["symmetric", "upper"])
@pytest.mark.parametrize("K, sizes, kind, ld_dtype", [
    (4, [700, 1400, 1700, 2400], "longrange", np.float32),
    (4, [3619, 333], "longrange", np.float32),
    (4, [1400, 1700], "sample", np.float32),
    (4, [1400, 2400], "longrange", np.int8),
    (10, [700, 1400, 2400], "longrange", np.float32),
    (20, [700, 1400, 2400], "longrange", np.float32),
    (4, [2400] * 45 + [2000] * 30 + [300] * 20, "longrange", np.int8),     # populous team classes (teams of 4, several blocks each)
], ids=lambda v: None if not isinstance(v, list) else "-".join(map(str, v[:3])) + (f"-x{len(v)}" if len(v) > 3 else ""))
def test_mixture_far_field(gpu, K, sizes, kind, ld_dtype, low_memory):
    from viprs_amd.vi import e_step_hip as S
    ld, ss, inp = syn.make_problem(sizes=sizes, low_memory=low_memory, ld_dtype=ld_dtype, seed=62, kind=kind)
    assert_far_field_matters(ld, inp)
    mix, st0 = _mixture_inputs(ld, ss, K)
    H.assert_state_equal(_run_mix(S, ld, inp, mix, st0, 2), _run_mix(O, ld, inp, mix, st0, 2))


def _run_grid(mod, ld, inp, g, st0, active, sweeps):
    st = {k: v.copy(order="F") for k, v in st0.items()}
    for _ in range(sweeps):
        mod.cpp_e_step_grid(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, inp.std_beta, st["var_gamma"], st["var_mu"],
                            st["eta"], st["q"], st["eta_diff"], g["u_logs"], g["hvt"], g["mu_mult"], ld.dq_scale,
                            active, 1, ld.low_memory)
    return st


@pytest.mark.parametrize("mfma", ["0", "1"], ids=["items", "mfma"])
@pytest.mark.parametrize("low_memory", [False, True], ids=["symmetric", "upper"])
@pytest.mark.parametrize("sizes, kind, ld_dtype, G, n_active", [
    ([700, 1400, 90], "longrange", np.float32, 32, 6),
    ([1700, 2400], "longrange", np.float32, 32, 5),
    ([3619], "longrange", np.float32, 8, 8),
    ([1400, 650], "sample", np.float32, 32, 4),
    ([1400, 1700], "longrange", np.int8, 12, 12),
])
def test_grid_far_field(gpu, sizes, kind, ld_dtype, G, n_active, low_memory, mfma, monkeypatch):
    from viprs_amd.vi import e_step_hip as S
    monkeypatch.setenv("VIPRS_GRID_MFMA", mfma)
    ld, ss, inp = syn.make_problem(sizes=sizes, low_memory=low_memory, ld_dtype=ld_dtype, seed=63, kind=kind)
    assert_far_field_matters(ld, inp)
    g, st0 = _grid_inputs(ld, ss, G)
    active = np.random.default_rng(G + n_active).permutation(G)[:n_active].astype(np.int32)
    H.assert_state_equal(_run_grid(S, ld, inp, g, st0, active, 2), _run_grid(O, ld, inp, g, st0, active, 2))


@pytest.mark.parametrize("low_memory", [False, True], ids=["symmetric", "upper"])
def test_grid_mfma_all_32_models_far_field(gpu, low_memory, monkeypatch):
    """All 32 columns of the batched kernel against the oracle (the oracle runs the 32 models one by one)."""
    from viprs_amd.vi import e_step_hip as S
    monkeypatch.setenv("VIPRS_GRID_MFMA", "1")
    ld, ss, inp = syn.make_problem(sizes=[1000, 1700, 300], low_memory=low_memory, seed=64, kind="longrange")
    g, st0 = _grid_inputs(ld, ss, 32)
    active = np.arange(32, dtype=np.int32)
    H.assert_state_equal(_run_grid(S, ld, inp, g, st0, active, 1), _run_grid(O, ld, inp, g, st0, active, 1))
