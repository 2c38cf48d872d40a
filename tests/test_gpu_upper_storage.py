"""GPU: the two storage forms of an upper-triangular plan's dense blocks.

The panel kernels and the batched grid kernel (resident / team blocks) sweep the MIRRORED form (the upper triangle
copied into the lower one: the reference's second pass, e_step.hpp:307-338 / :266-303, becomes coalesced strip updates
into per-row sums); the float64 kernels and the grid kernel's streaming form read the PACKED form (zeros on and left of
the diagonal).  A plan converts its blocks on its own stream when the kernel family of the coming launch asks for the
other form.  Whatever the order of the calls, every result is the reference's, bit for bit."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H
from tests.test_gpu_models import _run_grid, _run_mix
from tests.test_oracle_vs_ref import _grid_inputs, _mixture_inputs
from viprs_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ld_dtype", [np.float32, np.int8], ids=["f32", "int8"])
def test_one_plan_alternating_kernel_families(gpu, ld_dtype):
    """fp32 spike-and-slab (mirrored) -> float64 (packed) -> batched grid (mirrored) -> mixture (mirrored) -> float64 ->
    fp32 again, all on ONE plan (the plan cache keys on the LD arrays): `==` the oracle every time."""
    from viprs_amd.vi import e_step_hip as S
    sizes = [70, 1400, 333, 2000]                                   # single workgroups, a team block, a float64 big-class block
    ld, ss, inp = syn.make_problem(sizes=sizes, low_memory=True, ld_dtype=ld_dtype, seed=61, kind="longrange")
    _, _, inp64 = syn.make_problem(sizes=sizes, low_memory=True, ld_dtype=ld_dtype, seed=61, kind="longrange",
                                   float_precision=np.float64)
    g, st_g = _grid_inputs(ld, ss, 32)
    mix, st_m = _mixture_inputs(ld, ss, 4)
    active = np.arange(32, dtype=np.int32)
    S.clear_plan_cache()
    try:
        def f32():
            st0 = inp.state_copy()
            H.assert_state_equal(H.run_hip(ld, inp, st0, sweeps=2), H.run_oracle(ld, inp, st0, sweeps=2))

        def f64():
            st0 = inp64.state_copy()
            got = {k: v.copy() for k, v in st0.items()}
            ref = {k: v.copy() for k, v in st0.items()}
            for st, mod in ((got, S), (ref, O)):
                for _ in range(2):
                    mod.cpp_e_step(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, inp64.std_beta, st["var_gamma"], st["var_mu"],
                                   st["eta"], st["q"], st["eta_diff"], inp64.u_logs, inp64.sqrt_half_var_tau, inp64.mu_mult,
                                   ld.dq_scale, 1, True)
            H.assert_state_equal(got, ref)

        def grid():
            H.assert_state_equal(_run_grid(S, ld, inp, g, st_g, active), _run_grid(O, ld, inp, g, st_g, active))

        def mixture():
            H.assert_state_equal(_run_mix(S, ld, inp, mix, st_m, 2), _run_mix(O, ld, inp, mix, st_m, 2))

        for step in (f32, f64, grid, mixture, f64, f32, grid):
            step()
        plan = S.plan_for(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, True)
        assert plan.n_blocks == len(sizes)                           # (the cached plan that served all of it)
    finally:
        S.clear_plan_cache()


def test_mirrored_grid_with_a_partial_last_tile_and_few_models(gpu):
    """Batched grid kernel over mirrored blocks: block sizes around the 128-column tile / 64-row panel edges (the triangular
    half of a diagonal tile, the read-modify-write of q at the end of a block), a team block, 5 active models out of 7."""
    from viprs_amd.vi import e_step_hip as S
    ld, ss, inp = syn.make_problem(sizes=[1, 63, 64, 65, 127, 128, 129, 191, 193, 1537, 700], low_memory=True, ld_dtype=np.int16,
                                   seed=63, kind="longrange")
    g, st0 = _grid_inputs(ld, ss, 7)
    active = np.array([6, 0, 3, 4, 1], dtype=np.int32)
    got = _run_grid(S, ld, inp, g, st0, active, sweeps=3)
    ref = _run_grid(O, ld, inp, g, st0, active, sweeps=3)
    H.assert_state_equal(got, ref)
    cut = _run_grid(O, H.cut_far_field(ld), inp, g, st0, active, sweeps=3)
    assert np.max(np.abs(cut["q"] - ref["q"])) > 1e-4 * np.max(np.abs(ref["q"]))      # (the far field matters on this input)


def test_state_placement_probe_hands_out_a_fresh_state(gpu):
    """`DeviceState` of a large fp32 plan: several allocations are probed with a synthetic sweep and the fastest is kept
    (viprs_amd/plan.py, EXPERIMENTS.md round 5).  What comes back is a fresh state -- all zeros -- whose sweeps equal those of
    a state created without the probe, bit for bit."""
    from viprs_amd.plan import DeviceState, LDPlan
    sizes = [650] * 330                                            # 214 500 SNPs: above the probe's threshold
    ld, ss, inp = syn.make_problem(sizes=sizes, low_memory=True, ld_dtype=np.int8, seed=64, kind="longrange")
    plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, True)
    out = {}
    for mode in ("off", "probe"):
        ds = DeviceState(plan, "float32", "spike_slab", 1, placement=mode)
        if mode == "off":
            assert ds.placement is None
        else:
            p = ds.placement
            assert p["candidates"] == DeviceState.PLACEMENT_CANDIDATES and 0 <= p["chosen"] < p["candidates"]
            assert len(p["kernel_ms_min"]) == p["candidates"] and min(p["kernel_ms_min"]) > 0
            for name in ("std_beta", "u_logs", "mu_mult", "var_gamma", "eta", "q", "eta_diff"):
                assert not ds.download(name).any(), name           # handed out zeroed
        ds.upload("std_beta", inp.std_beta)
        for k in ("u_logs", "sqrt_half_var_tau", "mu_mult"):
            ds.upload(k, getattr(inp, k))
        ds.reset(inp.pi)
        for _ in range(2):
            ds.e_step(ld.dq_scale)
        out[mode] = {k: ds.download(k) for k in H.STATE}
        ds.close()
    H.assert_state_equal(out["probe"], out["off"])
    st0 = inp.state_copy()
    H.assert_state_equal(out["probe"], H.run_oracle(ld, inp, st0, sweeps=2))
    plan.close()
