"""One independent model per chromosome (the reference's default mode, bin/viprs_fit:232-238 + :1079-1086) fitted in lock
step on one plan (`VIPRSPerChromosome`): every chromosome's trajectory against fixtures made by fitting that chromosome
ALONE with the reference's own Python layer (tests/golden/make_fit_golden.py::per_chromosome_cases).

CPU: the host logic with the oracle's kernel through the `e_step_fn` test hook (+ a 2-rank gloo fit).
GPU: the batched fit against the fixtures, and `==` the same chromosomes fitted one after the other by `VIPRS`."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as O
from tests.test_fit import loader_from_fixture

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FITCHR = sorted(glob.glob(os.path.join(HERE, "golden", "fitchr_ss_*.npz")))      # (fitchr_mix_*: tests/test_per_chromosome_mix.py)
IDS = [os.path.basename(p)[:-4] for p in FITCHR]
FX4 = os.path.join(HERE, "golden", "fitchr_ss_4chr_upper.npz")     # four chromosomes, free sigma_epsilon


def model_kwargs(fx, e_step="oracle", **extra):
    kw = dict(low_memory=bool(fx["low_memory"]), dequantize_on_the_fly=bool(fx["dequantize_on_the_fly"]),
              float_precision=str(fx["float_precision"]), **extra)
    if not np.isnan(float(fx["fix_sigma_epsilon"])):
        kw["fix_params"] = {"sigma_epsilon": float(fx["fix_sigma_epsilon"])}
    if e_step == "oracle":
        kw["e_step_fn"] = O.cpp_e_step
    return kw


def theta_of(fx):
    return {"pi": float(fx["theta0_pi"]), "sigma_epsilon": float(fx["theta0_sigma_epsilon"])}


def build(fx, e_step="oracle", **extra):
    from viprs_amd.model import VIPRSPerChromosome
    return VIPRSPerChromosome(loader_from_fixture(fx), **model_kwargs(fx, e_step, **extra))


def check_against_fixture(model, fx, device_sums=False, fast=False):
    # Long-range LD is ill-conditioned: the reference's own posterior moves by up to 7e-4 per entry when ITS std_beta changes
    # by one ulp.  The device-resident iteration forms the M-step sums in float64 in a fixed order where the
    # reference calls np.sum, so after ~40 iterations single entries of pip sit ~1e-5 from the fixture (1 of 550 here).
    pip_atol = 2e-5 if (device_sums and str(fx["ld_kind"]) != "ar1") else 2e-6
    if fast and str(fx["ld_kind"]) != "ar1":
        pip_atol = 1e-4          # math_mode="fast" on that LD: still well inside the 7e-4 the reference itself moves by (one ulp)
    q = model.q_full if model.comm.world_size > 1 else model.q
    assert sorted(model.pip) == sorted(int(c) for c in fx["chroms"])
    for c in (int(c) for c in fx["chroms"]):
        h, ref = np.array(model.history[c]["ELBO"]), fx[f"elbo_history_{c}"]
        assert len(h) == len(ref), f"chromosome {c}: {len(h)} ELBO entries, the reference's own fit has {len(ref)}"
        np.testing.assert_allclose(h, ref, rtol=2e-7, atol=0.05)
        r = model.optim_results[c]
        # (math_mode="fast": the ELBO moves by ~5e-9 relative, and WHICH success rule fires first on the stopping iteration --
        #  ELBO within 1e-6 or max |eta_diff| < 1e-6 -- may differ; the iteration itself does not)
        assert (r.nit, r.success) == (int(fx[f"nit_{c}"]), bool(fx[f"success_{c}"]))
        assert fast or r.message == str(fx[f"message_{c}"])
        np.testing.assert_allclose(np.float64(model.pi[c]), fx[f"final_pi_{c}"], rtol=2e-4)
        np.testing.assert_allclose(np.float64(model.tau_beta[c]), fx[f"final_tau_beta_{c}"], rtol=2e-4)
        np.testing.assert_allclose(float(model.sigma_epsilon[c]), float(fx[f"final_sigma_epsilon_{c}"]), rtol=1e-5)
        np.testing.assert_allclose(float(model._sigma_g[c]), float(fx[f"final_sigma_g_{c}"]), rtol=1e-4)
        np.testing.assert_allclose(model.pip[c], fx[f"pip_{c}"], rtol=2e-3, atol=pip_atol)
        np.testing.assert_allclose(model.post_mean_beta[c], fx[f"post_mean_beta_{c}"], rtol=2e-3, atol=2e-7)
        np.testing.assert_allclose(q[c], fx[f"q_{c}"], rtol=2e-3, atol=2e-6)
        np.testing.assert_allclose(model.post_var_beta[c], fx[f"post_var_beta_{c}"], rtol=2e-3, atol=1e-9)
    # the chromosomes stop at different iterations: the fixtures exercise the convergence masks
    assert len({int(fx[f"nit_{int(c)}"]) for c in fx["chroms"]}) > 1


def sequential_fits(fx, e_step="oracle", **extra):
    """What the batched fit replaces: one `VIPRS` per chromosome on its own loader, one after the other."""
    from viprs_amd.model import VIPRS
    out = {}
    for c, sub in loader_from_fixture(fx).split_by_chromosome().items():
        out[c] = VIPRS(sub, **model_kwargs(fx, e_step, **extra)).fit(max_iter=100, theta_0=theta_of(fx))
    return out


def check_identical_to_sequential(model, seq):
    for c, one in seq.items():
        assert np.array_equal(model.history[c]["ELBO"], one.history["ELBO"], equal_nan=True), f"chromosome {c}: ELBO trajectories differ"
        r, r1 = model.optim_results[c], one.optim_result
        assert (r.nit, r.success, r.message) == (r1.nit, r1.success, r1.message)
        for name in ("pi", "tau_beta", "sigma_epsilon", "_sigma_g"):
            a, b = getattr(model, name)[c], getattr(one, name)
            assert a == b and np.asarray(a).dtype == np.asarray(b).dtype, (c, name, a, b)
        for name in ("pip", "post_mean_beta", "post_var_beta", "q", "var_gamma", "var_mu", "eta", "eta_diff", "var_tau"):
            assert np.array_equal(getattr(model, name)[c], getattr(one, name)[c]), (c, name)


def test_fixtures_present():
    assert len(FITCHR) >= 3


@pytest.mark.parametrize("path", FITCHR, ids=IDS)
def test_lockstep_fit_cpu_host_logic(path):
    fx = np.load(path)
    model = build(fx).fit(max_iter=100, theta_0=theta_of(fx))
    check_against_fixture(model, fx)
    # ... and it is, bit for bit, what one VIPRS per chromosome computes with the same kernel
    check_identical_to_sequential(model, sequential_fits(fx))
    t = model.to_theta_table()
    assert sorted(set(t["Chromosome"])) == sorted(int(c) for c in fx["chroms"])
    assert len(model.to_history_table()) == sum(len(h["ELBO"]) for h in model.history.values())


def test_per_chromosome_theta_and_tracked_params():
    fx = np.load(FX4)
    chroms = [int(c) for c in fx["chroms"]]
    model = build(fx, tracked_params=["pi", "sigma_epsilon", "heritability", "max_eta_diff"])
    theta = {c: dict(theta_of(fx)) for c in chroms}
    theta[chroms[0]]["pi"] = 0.02
    model.fit(max_iter=100, theta_0=theta)
    ref = build(fx).fit(max_iter=100, theta_0=theta_of(fx))
    assert model.history[chroms[0]]["ELBO"] != ref.history[chroms[0]]["ELBO"]
    for c in chroms[1:]:                           # the other chromosomes' models do not see that change
        assert model.history[c]["ELBO"] == ref.history[c]["ELBO"]
    for c in chroms:
        h = model.history[c]
        assert len(h["pi"]) == len(h["ELBO"]) == len(h["sigma_epsilon"]) == len(h["heritability"])
        assert h["pi"][-1] == model.pi[c]
    assert set(model.get_heritability()) == set(chroms)


def test_unsupported_arguments():
    fx = np.load(FX4)
    with pytest.raises(NotImplementedError):
        build(fx).fit(max_iter=3, theta_0=theta_of(fx), continued=True)


def test_negative_mse_restarts_only_that_chromosome():
    """VIPRS.py:1025-1037: a model whose MSE turns negative starts again with sigma_epsilon = 0.95 fixed -- here for one
    chromosome of the batch (its marginal effects blown up), the others unaffected; same trajectories as the serial fits."""
    from viprs_amd.data import ArrayDataLoader, SumstatsArrays
    from viprs_amd.model import VIPRS, VIPRSPerChromosome
    fx = np.load(FX4)
    gdl = loader_from_fixture(fx)
    bad = int(fx["chroms"][1])
    ss = dict(gdl.sumstats_table)
    ss[bad] = SumstatsArrays(ss[bad].get_snp_pseudo_corr() * np.float32(5.0), ss[bad].n_per_snp)
    gdl = ArrayDataLoader(gdl.ld, ss)
    kw = model_kwargs(fx)
    model = VIPRSPerChromosome(gdl, **kw).fit(max_iter=40, theta_0=theta_of(fx))
    seq = {c: VIPRS(sub, **model_kwargs(fx)).fit(max_iter=40, theta_0=theta_of(fx)) for c, sub in gdl.split_by_chromosome().items()}
    assert seq[bad].fix_params.get("sigma_epsilon") == 0.95, "the test input no longer triggers the restart"
    check_identical_to_sequential(model, seq)
    assert model.sigma_epsilon[bad] == 0.95


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
import torch.distributed as dist
dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
from tests.test_per_chromosome import build, check_against_fixture, theta_of
from tests.comm_torch import TorchDistComm
fx = np.load({path!r})
comm = TorchDistComm()
model = build(fx, comm=comm)
m_local = sum(model.shapes.values())
assert 0 < m_local < model.n_snps, (m_local, model.n_snps)
model.fit(max_iter=100, theta_0=theta_of(fx))
check_against_fixture(model, fx)
dist.barrier(); dist.destroy_process_group()
print("RANK_OK", sys.argv[1])
"""


def test_two_rank_gloo_lockstep_fit(tmp_path):
    """LD blocks of ALL chromosomes dealt to 2 ranks: every chromosome's sums are all-rank sums (one exchange per
    iteration for all groups) and the trajectories are those of the single-process fit."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, port=port, path=FX4))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {r}" in o, o[-3000:]


# ---- GPU ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("path", FITCHR, ids=IDS)
def test_lockstep_fit_hip(gpu, path):
    fx = np.load(path)
    model = build(fx, e_step="hip").fit(max_iter=100, theta_0=theta_of(fx))
    # one plan, one sweep per EM round -- and bit for bit what one device fit per chromosome computes
    assert list(model._plans) == ["*"]
    check_identical_to_sequential(model, sequential_fits(fx, e_step="hip"))
    check_against_fixture(model, fx, device_sums=True)


@pytest.mark.gpu
def test_group_prep_and_sums_equal_a_plan_per_chromosome(gpu):
    """C ABI: `viprs_state_prep_groups` / `viprs_state_sums_groups_*` on the merged plan against `viprs_state_prep` /
    `viprs_state_sums` on a plan that holds one chromosome only: `==` on every input vector and every sum; an inactive
    chromosome's blocks are not swept (`viprs_plan_set_active_blocks`)."""
    from viprs_amd.data import merge_ld_arrays
    from viprs_amd.plan import DeviceState, LDPlan
    from viprs_amd.utils import synthetic as syn
    sizes = {1: [700, 90, 1800], 2: [300], 3: [64, 65, 2400, 130]}
    lds = {c: syn.make_ld(s, low_memory=True, seed=20 + c, kind="longrange") for c, s in sizes.items()}
    sss = {c: syn.make_sumstats(lds[c], n=5e4 * c, seed=30 + c) for c in sizes}
    chroms = sorted(sizes)
    shapes = {c: lds[c].m for c in chroms}
    lb, ip, data, seg = merge_ld_arrays(chroms, shapes, {c: lds[c].ld_left_bound for c in chroms},
                                        {c: lds[c].ld_indptr for c in chroms}, {c: lds[c].ld_data for c in chroms})
    plan = LDPlan(lb, ip, data, True)
    st = DeviceState(plan, "float32", "spike_slab")
    st.upload("std_beta", np.concatenate([sss[c].std_beta for c in chroms]))
    st.set_n_per_snp(np.concatenate([sss[c].n_per_snp for c in chroms]))
    gs = np.array([0] + [seg[c][1] for c in chroms], dtype=np.int64)
    st.set_groups(gs)
    with pytest.raises(ValueError, match="cuts through"):
        st.set_groups(np.array([0, 100, gs[-1]], dtype=np.int64))
    st.set_groups(gs)
    hyper = {1: (0.01, 0.8, 900.0), 2: (0.02, 0.7, 300.0), 3: (0.005, 0.9, 2500.0)}          # pi, sigma_eps, tau_beta
    rows = np.array([[g, np.log(p) - np.log(1 - p), np.log(t), s, t, 1.0] for g, (p, s, t) in
                     ((g, hyper[c]) for g, c in enumerate(chroms))])
    init = np.concatenate([np.full(shapes[c], hyper[c][0], np.float32) for c in chroms])
    zeros = np.zeros(plan.m, np.float32)
    for name, a in (("var_gamma", init), ("var_mu", zeros), ("eta", zeros), ("q", zeros), ("eta_diff", zeros)):
        st.upload(name, a)
    st.prep_groups(rows)
    st.e_step(1.0)
    st.e_step(1.0)
    st.sums_groups_begin(np.arange(3), 1.0)
    got = st.sums_groups_end()
    after2 = {n: st.download(n) for n in ("var_gamma", "var_mu", "eta", "q", "eta_diff")}
    for g, c in enumerate(chroms):
        p1 = LDPlan(lds[c].ld_left_bound, lds[c].ld_indptr, lds[c].ld_data, True)
        s1 = DeviceState(p1, "float32", "spike_slab")
        s1.upload("std_beta", sss[c].std_beta)
        s1.set_n_per_snp(sss[c].n_per_snp)
        s1.reset(hyper[c][0])
        s1.prep(*rows[g, 1:])
        s1.e_step(1.0)
        s1.e_step(1.0)
        a, b = seg[c]
        for name in ("u_logs", "sqrt_half_var_tau", "mu_mult", "var_gamma", "var_mu", "eta", "q", "eta_diff"):
            full = after2[name] if name in after2 else st.download(name)
            assert np.array_equal(full[a:b], s1.download(name)), (c, name)
        assert np.array_equal(got[g], s1.sums(1.0)), (c, got[g], s1.sums(1.0))
    # chromosome 2 converged: its blocks leave the sweep, its state stays put; the others move on exactly as before
    starts, _ = plan.blocks()
    grp = np.searchsorted(gs, starts[:-1], side="right") - 1
    plan.set_active_blocks(grp != 1)
    st.e_step(1.0)
    after3 = {n: st.download(n) for n in after2}
    a, b = seg[2]
    for n in after2:
        assert np.array_equal(after3[n][a:b], after2[n][a:b]), n
        assert not np.array_equal(after3[n][:a], after2[n][:a]) or n == "var_gamma"
    plan.set_active_blocks(None)
    st2 = DeviceState(plan, "float32", "spike_slab")
    st2.upload("std_beta", np.concatenate([sss[c].std_beta for c in chroms]))
    st2.set_n_per_snp(np.concatenate([sss[c].n_per_snp for c in chroms]))
    st2.set_groups(gs)
    for name in after2:
        st2.upload(name, after2[name])
    st2.prep_groups(rows)
    st2.e_step(1.0)                                  # all blocks again
    ref3 = {n: st2.download(n) for n in after2}
    for n in after2:
        assert np.array_equal(after3[n][:a], ref3[n][:a]) and np.array_equal(after3[n][b:], ref3[n][b:]), n
    # a subset of groups: only the listed rows come back, in the order asked for
    st.sums_groups_begin(np.array([2, 0]), 1.0)
    sub = st.sums_groups_end()
    st.sums_groups_begin(np.arange(3), 1.0)
    full = st.sums_groups_end()
    assert np.array_equal(sub, full[[2, 0]])
