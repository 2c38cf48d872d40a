"""One independent ``VIPRSMix`` model per chromosome in lock step on one plan (`VIPRSMixPerChromosome`): every chromosome's
trajectory against a fixture made by fitting that chromosome ALONE with the reference's own Python layer
(tests/golden/make_fit_golden.py::per_chromosome_cases, `fitchr_mix_*`), and `==` the chromosomes fitted one after the other.

CPU: the host logic with the oracle's kernel through the `e_step_fn` test hook (+ a 2-rank gloo fit).
GPU: the batched fit on the device; the C ABI's group prep / sums of a mixture state against a plan per chromosome."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as O
from tests.test_fit import loader_from_fixture

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FXM = os.path.join(HERE, "golden", "fitchr_mix_k4_3chr_upper.npz")


def model_kwargs(fx, e_step="oracle", **extra):
    kw = dict(K=int(fx["K"]), low_memory=bool(fx["low_memory"]), dequantize_on_the_fly=bool(fx["dequantize_on_the_fly"]),
              float_precision=str(fx["float_precision"]), **extra)
    if e_step == "oracle":
        kw["e_step_fn"] = O.cpp_e_step_mixture
    return kw


def theta_of(fx):
    return {"pis": np.array(fx["theta0_pis"]), "sigma_epsilon": float(fx["theta0_sigma_epsilon"])}


def build(fx, e_step="oracle", **extra):
    from viprs_amd.model import VIPRSMixPerChromosome
    return VIPRSMixPerChromosome(loader_from_fixture(fx), **model_kwargs(fx, e_step, **extra))


def sequential_fits(fx, e_step="oracle", loader=None, max_iter=100, **extra):
    from viprs_amd.model import VIPRSMix
    gdl = loader if loader is not None else loader_from_fixture(fx)
    return {c: VIPRSMix(sub, **model_kwargs(fx, e_step, **extra)).fit(max_iter=max_iter, theta_0=theta_of(fx))
            for c, sub in gdl.split_by_chromosome().items()}


def check_against_fixture(model, fx, device_sums=False, var_atol=None):
    # post_var_beta = zeta - eta^2 cancels: sums formed in another order (device float64 sums, several ranks) move one entry of
    # 440 by 1.6e-8 (2.3e-3 of its value)
    var_atol = var_atol if var_atol is not None else (5e-8 if device_sums else 1e-9)
    q = model.q_full if model.comm.world_size > 1 else model.q
    for c in (int(c) for c in fx["chroms"]):
        h, ref = np.array(model.history[c]["ELBO"]), fx[f"elbo_history_{c}"]
        assert len(h) == len(ref), f"chromosome {c}: {len(h)} ELBO entries, the reference's own fit has {len(ref)}"
        np.testing.assert_allclose(h, ref, rtol=2e-7, atol=0.05)
        r = model.optim_results[c]
        assert (r.nit, r.success, r.message) == (int(fx[f"nit_{c}"]), bool(fx[f"success_{c}"]), str(fx[f"message_{c}"]))
        np.testing.assert_allclose(np.float64(model.pi[c]), fx[f"final_pi_{c}"], rtol=2e-3, atol=1e-8)
        np.testing.assert_allclose(np.float64(model.tau_beta[c]), fx[f"final_tau_beta_{c}"], rtol=2e-4)
        np.testing.assert_allclose(float(model.sigma_epsilon[c]), float(fx[f"final_sigma_epsilon_{c}"]), rtol=1e-5)
        np.testing.assert_allclose(float(model._sigma_g[c]), float(fx[f"final_sigma_g_{c}"]), rtol=1e-4)
        np.testing.assert_allclose(model.pip[c], fx[f"pip_{c}"], rtol=2e-3, atol=2e-5 if device_sums else 2e-6)
        np.testing.assert_allclose(model.post_mean_beta[c], fx[f"post_mean_beta_{c}"], rtol=2e-3, atol=2e-7)
        np.testing.assert_allclose(q[c], fx[f"q_{c}"], rtol=2e-3, atol=2e-6)
        np.testing.assert_allclose(model.post_var_beta[c], fx[f"post_var_beta_{c}"], rtol=2e-3, atol=var_atol)
    assert len({int(fx[f"nit_{int(c)}"]) for c in fx["chroms"]}) > 1          # the convergence masks are exercised


def check_identical_to_sequential(model, seq):
    for c, one in seq.items():
        assert np.array_equal(model.history[c]["ELBO"], one.history["ELBO"], equal_nan=True), f"chromosome {c}: ELBO trajectories differ"
        r, r1 = model.optim_results[c], one.optim_result
        assert (r.nit, r.success, r.message) == (r1.nit, r1.success, r1.message)
        for name in ("pi", "tau_beta", "sigma_epsilon", "_sigma_g"):
            a, b = getattr(model, name)[c], getattr(one, name)
            assert np.array_equal(a, b) and np.asarray(a).dtype == np.asarray(b).dtype, (c, name, a, b)
        for name in ("pip", "post_mean_beta", "post_var_beta", "q", "var_gamma", "var_mu", "eta", "eta_diff", "var_tau"):
            assert np.array_equal(getattr(model, name)[c], getattr(one, name)[c]), (c, name)


@pytest.mark.parametrize("host", ["vector", "scalar"])
def test_mix_lockstep_fit_cpu_host_logic(host):
    """Both forms of the host side: the array form (`LockstepMixEM`, the default) and `VIPRSMix`'s own scalar code per model."""
    fx = np.load(FXM)
    model = build(fx, host=host).fit(max_iter=100, theta_0=theta_of(fx))
    check_against_fixture(model, fx)
    check_identical_to_sequential(model, sequential_fits(fx))
    chroms = sorted(int(c) for c in fx["chroms"])
    t = model.to_theta_table()
    assert sorted(set(t["Chromosome"])) == chroms and {"pi_1", "pi_4", "tau_beta_4"} <= set(t["Parameter"])
    assert len(model.to_history_table()) == sum(len(h["ELBO"]) for h in model.history.values())
    assert set(model.get_heritability()) == set(model.get_proportion_causal()) == set(chroms)
    for c in chroms:
        assert model.pi[c].shape == (4,) and model.get_proportion_causal()[c] == np.sum(model.pi[c])
    # a second fit on the same object starts from scalars again
    again = model.fit(max_iter=100, theta_0=theta_of(fx))
    check_against_fixture(again, fx)


@pytest.mark.parametrize("precision", ["float32", "float64"])
@pytest.mark.parametrize("fix", [{}, {"sigma_epsilon": 0.85}, {"pi": 0.02}, {"tau_beta": 400.0}])
def test_mix_vector_host_equals_scalar_host(fix, precision):
    """The array form against the scalar form, `==` on everything a fit returns, with hyper-parameters fixed in the ways that
    change the DTYPES the serial code computes in (a fixed sigma_epsilon stays a float32 scalar, a fixed scalar tau_beta makes
    the starting tau vector float32, a fixed overall proportion rescales pi) and tracked parameters."""
    fx = np.load(FXM)
    theta = theta_of(fx)
    if "sigma_epsilon" in fix:
        theta.pop("sigma_epsilon")
    from viprs_amd.model import VIPRSMixPerChromosome
    out = {}
    for host in ("vector", "scalar"):
        kw = model_kwargs(fx, host=host, fix_params=dict(fix),
                          tracked_params=["pi", "heritability", "sigma_epsilon", "tau_beta", "sigma_g", "max_eta_diff"])
        kw["float_precision"] = precision
        out[host] = VIPRSMixPerChromosome(loader_from_fixture(fx), **kw).fit(max_iter=25, theta_0=dict(theta))
    a, b = out["vector"], out["scalar"]
    for c in a.groups:
        for key in a.history[c]:
            x, y = a.history[c][key], b.history[c][key]
            assert len(x) == len(y) and all(np.array_equal(u, v, equal_nan=True) and np.asarray(u).dtype == np.asarray(v).dtype
                                            for u, v in zip(x, y)), (c, key)
        r, r1 = a.optim_results[c], b.optim_results[c]
        assert (r.nit, r.success, r.message) == (r1.nit, r1.success, r1.message)
        for name in ("pi", "tau_beta", "sigma_epsilon", "_sigma_g"):
            u, v = getattr(a, name)[c], getattr(b, name)[c]
            assert np.array_equal(u, v) and np.asarray(u).dtype == np.asarray(v).dtype, (c, name, u, v)
        for name in ("pip", "post_mean_beta", "post_var_beta", "q", "var_gamma", "var_mu", "eta", "eta_diff", "var_tau"):
            assert np.array_equal(getattr(a, name)[c], getattr(b, name)[c]), (c, name)


def test_mix_per_chromosome_theta_and_tracked_params():
    fx = np.load(FXM)
    chroms = [int(c) for c in fx["chroms"]]
    theta = {c: dict(theta_of(fx)) for c in chroms}
    theta[chroms[0]]["pis"] = 2.0 * theta[chroms[0]]["pis"]
    model = build(fx, tracked_params=["pi", "sigma_epsilon", "heritability", "max_eta_diff"]).fit(max_iter=100, theta_0=theta)
    ref = build(fx).fit(max_iter=100, theta_0=theta_of(fx))
    assert model.history[chroms[0]]["ELBO"] != ref.history[chroms[0]]["ELBO"]
    for c in chroms[1:]:                           # the other chromosomes' models do not see that change
        assert model.history[c]["ELBO"] == ref.history[c]["ELBO"]
    for c in chroms:
        h = model.history[c]
        assert len(h["pi"]) == len(h["ELBO"]) == len(h["sigma_epsilon"]) == len(h["heritability"]) == len(h["max_eta_diff"])
        assert h["pi"][-1] == np.sum(model.pi[c])
    with pytest.raises(NotImplementedError):
        build(fx).fit(max_iter=3, theta_0=theta_of(fx), continued=True)


def test_mix_negative_mse_restarts_only_that_chromosome():
    """VIPRS.py:1025-1037 for one chromosome of the batch (its marginal effects blown up): same trajectories as the serial
    fits, the others unaffected."""
    from viprs_amd.data import ArrayDataLoader, SumstatsArrays
    from viprs_amd.model import VIPRSMixPerChromosome
    fx = np.load(FXM)
    gdl = loader_from_fixture(fx)
    bad = int(fx["chroms"][1])
    ss = dict(gdl.sumstats_table)
    ss[bad] = SumstatsArrays(ss[bad].get_snp_pseudo_corr() * np.float32(6.0), ss[bad].n_per_snp)
    gdl = ArrayDataLoader(gdl.ld, ss)
    seq = sequential_fits(fx, loader=gdl, max_iter=30)
    assert seq[bad].fix_params.get("sigma_epsilon") == 0.95, "the test input no longer triggers the restart"
    for host in ("vector", "scalar"):
        model = VIPRSMixPerChromosome(gdl, host=host, **model_kwargs(fx)).fit(max_iter=30, theta_0=theta_of(fx))
        check_identical_to_sequential(model, seq)
        assert model.sigma_epsilon[bad] == 0.95


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
import torch.distributed as dist
dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
from tests.test_per_chromosome_mix import build, check_against_fixture, theta_of
from tests.comm_torch import TorchDistComm
fx = np.load({path!r})
model = build(fx, comm=TorchDistComm())
m_local = sum(model.shapes.values())
assert 0 < m_local < int(model.gdl.m), (m_local, model.gdl.m)
model.fit(max_iter=100, theta_0=theta_of(fx))
# (sums added in another order across the ranks: post_var_beta = zeta - eta^2 cancels, one entry of 440 moves by 1.5e-8)
check_against_fixture(model, fx, var_atol=5e-8)
dist.barrier(); dist.destroy_process_group()
print("RANK_OK", sys.argv[1])
"""


def test_mix_two_rank_gloo_lockstep_fit(tmp_path):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, port=port, path=FXM))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {r}" in o, o[-3000:]


# ---- GPU ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_mix_lockstep_fit_hip(gpu):
    fx = np.load(FXM)
    model = build(fx, e_step="hip").fit(max_iter=100, theta_0=theta_of(fx))
    assert list(model._plans) == ["*"]
    check_identical_to_sequential(model, sequential_fits(fx, e_step="hip"))
    check_against_fixture(model, fx, device_sums=True)


@pytest.mark.gpu
@pytest.mark.parametrize("ftype", ["float32", "float64"])
def test_mixture_group_prep_and_sums_equal_a_plan_per_chromosome(gpu, ftype):
    """C ABI: `viprs_state_prep_mixture_groups` / `viprs_state_sums_mixture_groups_*` on the merged plan against
    `viprs_state_prep_mixture` / `viprs_state_sums_mixture_*` on a plan that holds one chromosome only: `==` on every input
    array and every sum."""
    from viprs_amd.data import merge_ld_arrays
    from viprs_amd.plan import DeviceState, LDPlan
    from viprs_amd.utils import synthetic as syn
    K, T = 4, np.dtype(ftype)
    sizes = {1: [700, 90, 1500], 2: [300], 3: [64, 65, 1900, 130]}
    lds = {c: syn.make_ld(s, low_memory=True, seed=40 + c, kind="longrange") for c, s in sizes.items()}
    sss = {c: syn.make_sumstats(lds[c], n=5e4 * c, seed=50 + c) for c in sizes}
    chroms = sorted(sizes)
    shapes = {c: lds[c].m for c in chroms}
    lb, ip, data, seg = merge_ld_arrays(chroms, shapes, {c: lds[c].ld_left_bound for c in chroms},
                                        {c: lds[c].ld_indptr for c in chroms}, {c: lds[c].ld_data for c in chroms})
    plan = LDPlan(lb, ip, data, True)
    st = DeviceState(plan, ftype, "mixture", K)
    st.upload("std_beta", np.concatenate([sss[c].std_beta for c in chroms]).astype(T))
    st.set_n_per_snp(np.concatenate([sss[c].n_per_snp for c in chroms]))
    gs = np.array([0] + [seg[c][1] for c in chroms], dtype=np.int64)
    st.set_groups(gs)
    rng = np.random.default_rng(7)
    hyper = {}
    for c in chroms:                                               # pis, tau_betas, sigma_eps
        hyper[c] = (0.01 * c * np.array([0.4, 0.3, 0.2, 0.1]), 500.0 * c * 2.0 ** np.arange(-3, 1), 0.9 - 0.1 * c)
    rows = np.array([np.concatenate([[g, np.log(1.0 - p.sum()), s, 1.0], np.log(p) - np.log(1.0 - p), np.log(t), t])
                     for g, (p, t, s) in ((g, hyper[c]) for g, c in enumerate(chroms))])
    init = np.concatenate([np.tile(hyper[c][0], (shapes[c], 1)) for c in chroms]).astype(T)
    zw, zv = np.zeros((plan.m, K), T), np.zeros(plan.m, T)
    lvt = rng.normal(size=(plan.m, K))
    for name, a in (("var_gamma", init), ("var_mu", zw), ("eta", zv), ("q", zv), ("eta_diff", zv)):
        st.upload(name, a)
    st.set_log_var_tau(lvt)
    st.prep_mixture_groups(rows)
    st.e_step(1.0)
    st.e_step(1.0)
    st.sums_mixture_groups_begin(np.arange(3), 1.0)
    got = st.sums_mixture_groups_end()
    assert got.shape == (3, 7 + 6 * K)
    full = {n: st.download(n) for n in ("u_logs", "sqrt_half_var_tau", "mu_mult", "log_null_pi", "var_gamma", "var_mu", "eta", "q",
                                        "eta_diff")}
    for g, c in enumerate(chroms):
        p1 = LDPlan(lds[c].ld_left_bound, lds[c].ld_indptr, lds[c].ld_data, True)
        s1 = DeviceState(p1, ftype, "mixture", K)
        a, b = seg[c]
        s1.upload("std_beta", sss[c].std_beta.astype(T))
        s1.set_n_per_snp(sss[c].n_per_snp)
        for name, arr in (("var_gamma", init[a:b]), ("var_mu", zw[a:b]), ("eta", zv[a:b]), ("q", zv[a:b]), ("eta_diff", zv[a:b])):
            s1.upload(name, np.ascontiguousarray(arr))
        s1.set_log_var_tau(lvt[a:b])
        pis, taus, sig = hyper[c]
        s1.prep_mixture(np.log(pis) - np.log(1.0 - pis), np.log(taus), taus, np.log(1.0 - pis.sum()), sig, 1.0)
        s1.e_step(1.0)
        s1.e_step(1.0)
        for name, arr in full.items():
            assert np.array_equal(arr[a:b], s1.download(name)), (c, name)
        s1.sums_mixture_begin(1.0)
        assert np.array_equal(got[g], s1.sums_mixture_end()), (c, got[g])
    # a subset of the groups, in the order asked for
    st.sums_mixture_groups_begin(np.array([2, 0]), 1.0)
    assert np.array_equal(st.sums_mixture_groups_end(), got[[2, 0]])
    # error behaviour: the spike-and-slab entry points refuse a mixture state (and the other way round), a row that names no
    # group, sums before any prep of a fresh state, groups on a mixture wider than the device-resident iteration covers
    with pytest.raises(ValueError):
        st.prep_groups(np.zeros((1, 6)))
    with pytest.raises(ValueError):
        st.sums_groups_begin(np.arange(3), 1.0)
    bad = rows[:1].copy()
    bad[0, 0] = 3
    with pytest.raises(ValueError, match="out of range"):
        st.prep_mixture_groups(bad)
    ss_state = DeviceState(plan, ftype, "spike_slab")
    ss_state.set_groups(gs)
    with pytest.raises(ValueError):
        ss_state.prep_mixture_groups(np.zeros((1, 7)))
    fresh = DeviceState(plan, ftype, "mixture", K)
    fresh.set_groups(gs)
    with pytest.raises(ValueError, match="have not been called"):
        fresh.sums_mixture_groups_begin(np.arange(3), 1.0)
    wide = DeviceState(plan, ftype, "mixture", 10)
    with pytest.raises(NotImplementedError):
        wide.set_groups(gs)
