"""GPU: mixture and grid E-steps (panel kernels with the model policies) against the oracle."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H
from tests.test_oracle_vs_ref import _grid_inputs, _mixture_inputs
from viprs_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu
STATE = ("var_gamma", "var_mu", "eta", "q", "eta_diff")
# long-range, non-Toeplitz LD: every entry of a block changes the result (synthetic.py; AR(1) blocks are blind
# to everything further than ~2 panels from the diagonal)
KIND = "longrange"


def _run_mix(mod, ld, inp, mix, st0, sweeps, **kw):
    st = {k: v.copy() for k, v in st0.items()}
    for _ in range(sweeps):
        mod.cpp_e_step_mixture(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, inp.std_beta, st["var_gamma"], st["var_mu"],
                               st["eta"], st["q"], st["eta_diff"], mix["log_null_pi"], mix["u_logs"], mix["shvt"],
                               mix["mu_mult"], ld.dq_scale, 1, ld.low_memory, **kw)
    return st


@pytest.mark.parametrize("low_memory", [False, True])
# K <= 8: panel kernels, components across the lanes (ordered sums as scalar chains; the max by DPP unless K == 4); 9..31: the same with
# the component inputs streamed from global memory (10 = the reference's own test,
# 20 = its benchmark; 15 / 16 / 31: the edges of the two instantiations); 33: generic kernel
@pytest.mark.parametrize("K", [1, 4, 8, 9, 10, 15, 16, 20, 31, 33])
def test_mixture_matches_oracle(gpu, K, low_memory):
    from viprs_amd.vi import e_step_hip as S
    ld, ss, inp = syn.make_problem(sizes=[70, 1400, 333, 2400], low_memory=low_memory, seed=31, kind=KIND)   # single workgroups + a team block
    mix, st0 = _mixture_inputs(ld, ss, K)
    ref = _run_mix(O, ld, inp, mix, st0, 2)
    got = _run_mix(S, ld, inp, mix, st0, 2)
    for k in STATE:
        H.assert_close(got[k], ref[k], 1e-5, k)
    H.assert_state_equal(got, ref)     # bit-for-bit in both LD forms


@pytest.mark.parametrize("mfma", ["0", "1"])          # per-(block, model) panel items / batched matrix-core kernel
@pytest.mark.parametrize("low_memory", [False, True])
def test_grid_matches_oracle(gpu, low_memory, mfma, monkeypatch):
    from viprs_amd.vi import e_step_hip as S
    monkeypatch.setenv("VIPRS_GRID_MFMA", mfma)
    ld, ss, inp = syn.make_problem(sizes=[130, 1300, 64], low_memory=low_memory, seed=33, kind=KIND)
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
    H.assert_state_equal(out["hip"], out["ref"])     # bit-for-bit in both LD forms
    untouched = [c for c in range(32) if c not in active]
    assert np.all(out["hip"]["eta"][:, untouched] == 0)


def _run_grid(mod, ld, inp, g, st0, active, sweeps=2):
    st = {k: v.copy(order="F") for k, v in st0.items()}
    for _ in range(sweeps):
        mod.cpp_e_step_grid(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, inp.std_beta, st["var_gamma"], st["var_mu"],
                            st["eta"], st["q"], st["eta_diff"], g["u_logs"], g["hvt"], g["mu_mult"], ld.dq_scale,
                            active, 1, ld.low_memory)
    return st


@pytest.mark.parametrize("low_memory", [False, True])
@pytest.mark.parametrize("G, n_active", [(1, 1), (3, 3), (32, 32), (40, 40), (70, 37)])
def test_grid_mfma_model_counts(gpu, G, n_active, low_memory, monkeypatch):
    """Batched kernel: fewer than / exactly / more than 32 models (chunks of 32), scattered active lists."""
    from viprs_amd.vi import e_step_hip as S
    monkeypatch.setenv("VIPRS_GRID_MFMA", "1")
    ld, ss, inp = syn.make_problem(sizes=[200, 96, 333], low_memory=low_memory, seed=35, kind=KIND)
    g, st0 = _grid_inputs(ld, ss, G)
    active = np.random.default_rng(G).permutation(G)[:n_active].astype(np.int32)
    ref = _run_grid(O, ld, inp, g, st0, active)
    got = _run_grid(S, ld, inp, g, st0, active)
    for k in STATE:
        H.assert_close(got[k], ref[k], 1e-5, k)
    H.assert_state_equal(got, ref)     # bit-for-bit in both LD forms


@pytest.mark.parametrize("ld_dtype", [np.int8, np.int16])
@pytest.mark.parametrize("low_memory", [False, True])
def test_grid_mfma_quantised_ld(gpu, ld_dtype, low_memory, monkeypatch):
    from viprs_amd.vi import e_step_hip as S
    monkeypatch.setenv("VIPRS_GRID_MFMA", "1")
    ld, ss, inp = syn.make_problem(sizes=[150, 520, 77], low_memory=low_memory, ld_dtype=ld_dtype, seed=36, kind=KIND)
    g, st0 = _grid_inputs(ld, ss, 12)
    active = np.arange(12, dtype=np.int32)
    ref = _run_grid(O, ld, inp, g, st0, active)
    got = _run_grid(S, ld, inp, g, st0, active)
    for k in STATE:
        H.assert_close(got[k], ref[k], 1e-5, k)
    H.assert_state_equal(got, ref)     # bit-for-bit in both LD forms


def test_grid_mfma_block_shapes(gpu, monkeypatch):
    """Panel / tile boundaries of the batched kernel: one SNP, 63/64/65, odd and even numbers of panels,
    blocks that end inside a 128-column tile."""
    from viprs_amd.vi import e_step_hip as S
    monkeypatch.setenv("VIPRS_GRID_MFMA", "1")
    sizes = [1, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 2, 321, 448, 449]
    ld, ss, inp = syn.make_problem(sizes=sizes, low_memory=False, seed=37, kind=KIND)
    g, st0 = _grid_inputs(ld, ss, 9)
    active = np.array([8, 0, 3, 4, 1, 7], dtype=np.int32)
    ref = _run_grid(O, ld, inp, g, st0, active, sweeps=3)
    got = _run_grid(S, ld, inp, g, st0, active, sweeps=3)
    H.assert_state_equal(got, ref)


@pytest.mark.parametrize("ld_dtype", [np.float32, np.int8])
def test_grid_upper_batched_equals_item_path(gpu, ld_dtype, monkeypatch):
    """Upper-triangular form: the batched kernels (matrix-core sweep + matrix-core second pass) and the
    per-(block, model) panel items run the same fma chains -- their states must agree bit for bit."""
    from viprs_amd.vi import e_step_hip as S
    res = {}
    for mfma in ("0", "1"):
        monkeypatch.setenv("VIPRS_GRID_MFMA", mfma)
        ld, ss, inp = syn.make_problem(sizes=[257, 700, 31, 129], low_memory=True, ld_dtype=ld_dtype, seed=38, kind=KIND)
        g, st0 = _grid_inputs(ld, ss, 11)
        res[mfma] = _run_grid(S, ld, inp, g, st0, np.arange(11, dtype=np.int32), sweeps=3)
    H.assert_state_equal(res["1"], res["0"])


@pytest.mark.parametrize("low_memory", [False, True])
def test_very_large_blocks_all_models(gpu, low_memory, monkeypatch):
    """LD blocks far beyond BASELINE's largest (6 000 SNPs): 13 000- and 9 001-SNP blocks through the
    8-CU teams, the mixture teams and the batched grid kernel -- bit-for-bit against the oracle."""
    from viprs_amd.vi import e_step_hip as S
    ld, ss, inp = syn.make_problem(sizes=[13000, 9001, 65], low_memory=low_memory, seed=47, kind=KIND)
    st0 = inp.state_copy()
    H.assert_state_equal(H.run_hip(ld, inp, st0, sweeps=1), H.run_oracle(ld, inp, st0, sweeps=1))
    mix, mst0 = _mixture_inputs(ld, ss, 4)
    H.assert_state_equal(_run_mix(S, ld, inp, mix, mst0, 1), _run_mix(O, ld, inp, mix, mst0, 1))
    monkeypatch.setenv("VIPRS_GRID_MFMA", "1")
    g, gst0 = _grid_inputs(ld, ss, 8)
    active = np.arange(8, dtype=np.int32)
    H.assert_state_equal(_run_grid(S, ld, inp, g, gst0, active, sweeps=1), _run_grid(O, ld, inp, g, gst0, active, sweeps=1))


@pytest.mark.parametrize("low_memory", [False, True])
def test_grid_blocks_on_both_sides_of_the_resident_limit(gpu, low_memory, monkeypatch):
    """The batched grid kernel keeps q of a block in accumulator registers for the whole block: one workgroup for blocks of
    up to 1 536 SNPs, a team beyond -- `==` the oracle on blocks on both sides of the limit, partial model lists."""
    from viprs_amd.vi import e_step_hip as S
    monkeypatch.setenv("VIPRS_GRID_MFMA", "1")
    ld, ss, inp = syn.make_problem(sizes=[1536, 1537, 640, 65, 1], low_memory=low_memory, seed=71, kind=KIND)
    g, st0 = _grid_inputs(ld, ss, 32)
    active = np.array([0, 31, 6, 17, 23], dtype=np.int32)
    H.assert_state_equal(_run_grid(S, ld, inp, g, st0, active, sweeps=2), _run_grid(O, ld, inp, g, st0, active, sweeps=2))


@pytest.mark.parametrize("low_memory", [False, True], ids=["symmetric", "upper"])
@pytest.mark.parametrize("sizes, ld_dtype, G, n_active", [
    ([1537], np.float32, 32, 5),                   # one tile beyond the resident form: a team of 2, the second member holds one tile
    ([3072, 100], np.float32, 32, 4),              # exactly two members' worth of tiles
    ([3073, 700], np.float32, 8, 8),               # three members, the third holds one tile
    ([3619, 1600, 650], np.float32, 32, 6),        # cfg3's largest block + a team of 2 + a single-workgroup block
    ([4700], np.float32, 12, 3),                   # four members
    ([2400, 1700], np.int8, 32, 5),
    ([1700, 3000], np.int16, 6, 6),
])
def test_grid_mfma_team_blocks(gpu, sizes, ld_dtype, G, n_active, low_memory, monkeypatch):
    """Blocks beyond the batched grid kernel's resident form (> 1 536 SNPs) on TEAMS of workgroups with a migrating chain
    (estep_grid_mfma.h, GridTeam): `==` the oracle on far-field LD, two sweeps (the second one meets the first one's
    generation-tagged a-vector granules)."""
    from viprs_amd.vi import e_step_hip as S
    monkeypatch.setenv("VIPRS_GRID_MFMA", "1")
    ld, ss, inp = syn.make_problem(sizes=sizes, low_memory=low_memory, ld_dtype=ld_dtype, seed=41, kind="longrange")
    g, st0 = _grid_inputs(ld, ss, G)
    active = np.random.default_rng(G + n_active).permutation(G)[:n_active].astype(np.int32)
    ref = _run_grid(O, ld, inp, g, st0, active, sweeps=2)
    got = _run_grid(S, ld, inp, g, st0, active, sweeps=2)
    H.assert_state_equal(got, ref)


@pytest.mark.parametrize("low_memory", [False, True], ids=["symmetric", "upper"])
def test_grid_mfma_more_team_blocks_than_fit(gpu, monkeypatch, low_memory):
    """More blocks beyond the resident form than the chip has workgroups for at once (170 x 1 600-SNP blocks = 340 team
    workgroups on 256 CUs, 128 of them beside the queue in the first launch): the team blocks go out in SEVERAL launches, a
    chipful of teams at a time (launch_grid.inc) -- `==` the oracle on a sample of models, and bit-reproducible."""
    from viprs_amd.vi import e_step_hip as S
    monkeypatch.setenv("VIPRS_GRID_MFMA", "1")
    ld, ss, inp = syn.make_problem(sizes=[1700] + [1600] * 169 + [300] * 10, low_memory=low_memory, seed=43, kind=KIND)
    g, st0 = _grid_inputs(ld, ss, 8)
    active = np.arange(8, dtype=np.int32)
    got = _run_grid(S, ld, inp, g, st0, active, sweeps=2)
    again = _run_grid(S, ld, inp, g, st0, active, sweeps=2)
    H.assert_state_equal(got, again)
    # ... and two of the models against the oracle
    ref = _run_grid(O, ld, inp, g, st0, active[:2], sweeps=2)
    got2 = _run_grid(S, ld, inp, g, st0, active[:2], sweeps=2)
    H.assert_state_equal(got2, ref)
    for k in H.STATE:                                  # (the 8-model run holds the same two columns)
        assert np.array_equal(got[k][:, :2] if got[k].ndim == 2 else got[k], got2[k][:, :2] if got2[k].ndim == 2 else got2[k]), k
