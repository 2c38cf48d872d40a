"""EM loop (`VIPRS.fit / e_step / m_step / elbo`, `VIPRSMix`): trajectories against fixtures captured
from the reference's own Python layer (tests/golden/make_fit_golden.py).

CPU: the host logic is driven with the oracle's kernels through the `e_step_fn` test hook.
GPU: the same trajectories with the HIP E-step (the product path).
world_size-2 gloo: LD blocks sharded over two ranks, one float64 exchange per iteration."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as O
from viprs_amd.data import ArrayDataLoader, LDArrays, SumstatsArrays
from viprs_amd.utils import synthetic as syn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FIT = sorted(glob.glob(os.path.join(HERE, "golden", "fit_*.npz")))     # (fitgrid_*: tests/test_grid.py)


def loader_from_fixture(fx):
    ld, ss = {}, {}
    for c in fx["chroms"]:
        c = int(c)
        sizes, rho = fx[f"sizes_{c}"], fx[f"rho_{c}"]
        if f"ld_upper_data_{c}" in fx:
            # round-3 fixtures carry their LD (long-range blocks, int8, upper-triangular store); the symmetric
            # form is its mirror image
            from viprs_amd.data import mirror_upper_ld
            ip, data = fx[f"ld_upper_indptr_{c}"], fx[f"ld_upper_data_{c}"]
            m = ip.shape[0] - 1
            dq = 1.0 / np.iinfo(data.dtype).max if np.issubdtype(data.dtype, np.integer) else 1.0
            ld[c] = LDArrays(symmetric=mirror_upper_ld(ip, data), upper=(np.arange(1, m + 1, dtype=np.int32), ip, data),
                             stored_dtype=data.dtype, dq_scale=dq)
        else:
            sym = syn.make_ld(sizes, low_memory=False, rho=rho)
            up = syn.make_ld(sizes, low_memory=True, rho=rho)
            ld[c] = LDArrays(symmetric=(sym.ld_left_bound, sym.ld_indptr, sym.ld_data),
                             upper=(up.ld_left_bound, up.ld_indptr, up.ld_data))
        ss[c] = SumstatsArrays(fx[f"std_beta_{c}"], fx[f"n_per_snp_{c}"])
    return ArrayDataLoader(ld, ss, n=float(fx["n"]) if "n" in fx else None)


def build_model(fx, comm=None, e_step="oracle", math_mode="exact"):
    from viprs_amd.model import VIPRS, VIPRSMix
    K = int(fx["K"])
    kw = dict(low_memory=bool(fx["low_memory"]), comm=comm, math_mode=math_mode,
              dequantize_on_the_fly=bool(fx["dequantize_on_the_fly"]) if "dequantize_on_the_fly" in fx else False,
              float_precision=str(fx["float_precision"]) if "float_precision" in fx else "float32")
    if not np.isnan(float(fx["fix_sigma_epsilon"])):
        kw["fix_params"] = {"sigma_epsilon": float(fx["fix_sigma_epsilon"])}
    if e_step == "oracle":
        kw["e_step_fn"] = O.cpp_e_step_mixture if K else O.cpp_e_step
    gdl = loader_from_fixture(fx)
    if K:
        model, theta = VIPRSMix(gdl, K=K, **kw), {"pis": fx["theta0_pis"], "sigma_epsilon": float(fx["theta0_sigma_epsilon"])}
    else:
        model, theta = VIPRS(gdl, **kw), {"pi": float(fx["theta0_pi"]), "sigma_epsilon": float(fx["theta0_sigma_epsilon"])}
    return model, theta


def check_against_fixture(model, fx, local_only=False, pi_rtol=2e-4):
    h = np.array(model.history["ELBO"])
    ref = fx["elbo_history"]
    assert len(h) == len(ref), f"{len(h)} ELBO entries, reference has {len(ref)}"
    np.testing.assert_allclose(h, ref, rtol=2e-7, atol=0.05)
    assert model.optim_result.nit == int(fx["nit"])
    assert model.optim_result.success == bool(fx["success"])
    assert model.optim_result.message == str(fx["message"])
    np.testing.assert_allclose(np.asarray(model.pi, dtype=np.float64), fx["final_pi"], rtol=pi_rtol)
    np.testing.assert_allclose(np.asarray(model.tau_beta, dtype=np.float64), fx["final_tau_beta"], rtol=2e-4)
    np.testing.assert_allclose(float(model.sigma_epsilon), float(fx["final_sigma_epsilon"]), rtol=1e-5)
    np.testing.assert_allclose(float(model._sigma_g), float(fx["final_sigma_g"]), rtol=1e-4)
    q = model.q_full if model.comm.world_size > 1 else model.q     # several ranks: gathered at the end of fit()
    assert sorted(model.pip) == sorted(int(c) for c in fx["chroms"])
    for c in sorted(model.pip):
        np.testing.assert_allclose(model.pip[c], fx[f"pip_{c}"], rtol=2e-3, atol=2e-6)
        np.testing.assert_allclose(model.post_mean_beta[c], fx[f"post_mean_beta_{c}"], rtol=2e-3, atol=2e-7)
        np.testing.assert_allclose(q[c], fx[f"q_{c}"], rtol=2e-3, atol=2e-6)
        np.testing.assert_allclose(model.post_var_beta[c], fx[f"post_var_beta_{c}"], rtol=2e-3, atol=1e-9)
    # pseudo-validation against the marginal effects of a second cohort: the reference's own
    # BayesPRSModel.pseudo_validate() -> _streamlined_pseudo_r2 (BayesPRSModel.py:397-410, pseudo_metrics.py:130-152)
    model.validation_std_beta = {int(c): fx[f"validation_std_beta_{int(c)}"] for c in fx["chroms"]}
    np.testing.assert_allclose(float(model.pseudo_validate()), float(fx["pseudo_r2"]), rtol=1e-4)


def test_fit_fixtures_present():
    assert len(FIT) >= 10


@pytest.mark.parametrize("path", FIT, ids=[os.path.basename(p)[:-4] for p in FIT])
def test_fit_trajectory_cpu_host_logic(path):
    fx = np.load(path)
    model, theta = build_model(fx, e_step="oracle")
    model.fit(max_iter=60, theta_0=theta)
    check_against_fixture(model, fx)


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIT, ids=[os.path.basename(p)[:-4] for p in FIT])
def test_fit_trajectory_hip(gpu, path):
    fx = np.load(path)
    model, theta = build_model(fx, e_step="hip")
    model.fit(max_iter=60, theta_0=theta)
    # device-resident mixture iteration: sum_j gamma_jk is accumulated in float64 on the device, while the
    # reference (and the host path) sums the (m, K) float32 array row by row in float32 (error ~ m 2^-24,
    # visible on the smallest components of pi) -- everything else keeps the common tolerances
    check_against_fixture(model, fx, pi_rtol=2e-3 if int(fx["K"]) else 2e-4)


def test_continued_fit_and_warm_start():
    fx = np.load(os.path.join(HERE, "golden", "fit_ss_1chr_upper.npz"))
    model, theta = build_model(fx)
    model.fit(max_iter=4, theta_0=dict(theta))
    n1 = len(model.history["ELBO"])
    model.fit(max_iter=56, continued=True)
    assert len(model.history["ELBO"]) > n1
    np.testing.assert_allclose(model.history["ELBO"][-1], fx["elbo_history"][-1], rtol=2e-7)


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
import torch.distributed as dist
dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
from tests.test_fit import build_model, check_against_fixture
from tests.comm_torch import TorchDistComm
fx = np.load({path!r})
comm = TorchDistComm()
model, theta = build_model(fx, comm=comm)
# LD BLOCKS are the sharded unit: both ranks hold part of the SNPs (of every chromosome that has >= 2 blocks)
m_local = sum(model.shapes.values())
assert 0 < m_local < model.n_snps, (m_local, model.n_snps)
tot = comm.allreduce_sum(np.array([float(m_local)]))
assert int(tot[0]) == model.n_snps
if {random_theta}:
    theta = None                      # random initialisation: every rank must end up with rank 0's draw
    np.random.seed(1234 + comm.rank)
model.fit(max_iter=60, theta_0=theta)
if {random_theta}:
    v = comm.allreduce_max(np.array([float(model.history["ELBO"][0]), -float(model.history["ELBO"][0])]))
    assert v[0] == -v[1], "ranks started from different hyper-parameters"
else:
    # mixture: sum_j gamma_jk is a float32 row-order sum in the reference (error ~ m 2^-24 on the smallest
    # components of pi); sharding changes the summation order, as the device-resident float64 sums do
    check_against_fixture(model, fx, pi_rtol=2e-3 if int(fx["K"]) else 2e-4)
dist.barrier(); dist.destroy_process_group()
print("RANK_OK", sys.argv[1])
"""


@pytest.mark.parametrize("name,random_theta", [("fit_ss_2chr_upper", False), ("fit_ss_1chr_sym", False),
                                               ("fit_mix_k4_upper", False), ("fit_ss_1chr_upper", True)])
def test_two_rank_gloo_fit_matches_single_process(tmp_path, name, random_theta):
    """LD blocks sharded over 2 ranks (gloo on CPU), including single-chromosome fits; hyper-parameters /
    ELBO follow from ONE exchange of the partial sums per iteration and reproduce the single-process
    reference trajectory; the posterior of ALL SNPs is on every rank when fit() returns.  With a random
    start every rank takes rank 0's draw."""
    path = os.path.join(HERE, "golden", name + ".npz")
    script = tmp_path / "worker.py"
    port = 29500 + ((os.getpid() * 7 + len(name) * 131 + int(random_theta)) % 2000)
    script.write_text(_WORKER.format(root=ROOT, port=port, path=path, random_theta=random_theta))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {r}" in o, o[-3000:]


@pytest.mark.parametrize("name", ["fit_ss_2chr_upper", "fit_mix_k4_upper"])
def test_reporting_methods_match_their_definitions(name):
    """entropy / loglikelihood / log_prior (the ELBO's parts), theta and history tables: evaluated from the
    reduced partial sums, checked against a direct NumPy evaluation of VIPRS.py:583-687 on the state arrays."""
    fx = np.load(os.path.join(HERE, "golden", name + ".npz"))
    model, theta = build_model(fx)
    model.fit(max_iter=12, theta_0=dict(theta))
    res = np.finfo(np.float64).resolution
    cat = lambda d: np.concatenate([np.asarray(d[c], dtype=np.float64) for c in model.chromosomes])
    g = np.clip(cat(model.var_gamma), res, 1 - res)
    pip = cat(model.compute_pip())
    ng = np.clip(1.0 - pip, res, 1 - res)
    lvt = cat(model._log_var_tau)
    pi, tau = np.asarray(model.pi, dtype=np.float64), np.asarray(model.tau_beta, dtype=np.float64)
    m = model.n_snps
    ent = 0.5 * m * (np.log(2 * np.pi) + 1) - (g * np.log(g)).sum() - (ng * np.log(ng)).sum() - 0.5 * (g * lvt).sum()
    np.testing.assert_allclose(model.entropy(), ent, rtol=1e-9)
    eta, beta = cat(model.eta), cat(model.std_beta)
    ll = -0.5 * model.n * (np.log(2 * np.pi * model.sigma_epsilon)
                           + (1.0 / model.sigma_epsilon) * (1.0 - 2.0 * beta.dot(eta) + model._sigma_g))
    np.testing.assert_allclose(model.loglikelihood(), ll, rtol=1e-7)
    mu, vt = cat(model.var_mu), cat(model.var_tau)
    lp = 0.5 * (g * np.log(tau)).sum() + (g * np.log(pi)).sum() + (ng * np.log(model.get_null_pi())).sum()
    lp -= 0.5 * (g * tau * (mu ** 2 + 1.0 / vt)).sum() + 0.5 * m * np.log(2 * np.pi)
    np.testing.assert_allclose(model.log_prior(), lp, rtol=1e-5)
    np.testing.assert_allclose(model.complete_loglikelihood(), model.loglikelihood() + model.log_prior())
    tt = model.to_theta_table()
    assert list(tt.columns) == ["Parameter", "Value"] and {"ELBO", "Residual_variance", "Heritability"} <= set(tt["Parameter"])
    assert ("pi_1" in set(tt["Parameter"])) == bool(int(fx["K"]))
    ht = model.to_history_table()
    assert len(ht) == len(model.history["ELBO"]) and "ELBO" in ht.columns
    assert model.get_average_effect_size_variance() > 0


_WORKER_EMPTY = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
import torch.distributed as dist
world = {world}
dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=world)
from oracle import oracle as O
from viprs_amd.data import ArrayDataLoader
from viprs_amd.model import VIPRS
from tests.comm_torch import TorchDistComm
comm = TorchDistComm()
# two chromosomes with ONE and TWO LD blocks: with three ranks every rank gets one block, with four one rank gets NONE
gdl = ArrayDataLoader.synthetic({{21: [260], 22: [150, 330]}}, seed=77, kind="longrange")
theta = {{"pi": 0.02, "sigma_epsilon": 0.85}}
model = VIPRS(gdl, low_memory=True, comm=comm, e_step_fn=O.cpp_e_step)
n_local = sum(model.shapes.values())
tot = comm.allreduce_sum(np.array([float(n_local), 1.0 if n_local == 0 else 0.0]))
assert int(tot[0]) == 740 and int(tot[1]) == world - 3, (tot, n_local)
model.fit(max_iter=25, theta_0=dict(theta))
if comm.rank == 0:
    ref = VIPRS(gdl, low_memory=True, e_step_fn=O.cpp_e_step).fit(max_iter=25, theta_0=dict(theta))
    np.testing.assert_allclose(model.history["ELBO"], ref.history["ELBO"], rtol=2e-7, atol=0.05)
    assert model.optim_result.nit == ref.optim_result.nit
    for c in (21, 22):
        np.testing.assert_allclose(model.pip[c], ref.pip[c], rtol=2e-3, atol=2e-6)
        np.testing.assert_allclose(model.post_mean_beta[c], ref.post_mean_beta[c], rtol=2e-3, atol=2e-7)
        np.testing.assert_allclose(model.q_full[c], ref.q[c], rtol=2e-3, atol=2e-6)
        assert model.pip[c].shape == (gdl.shapes[c],)
dist.barrier(); dist.destroy_process_group()
print("RANK_OK", sys.argv[1])
"""


@pytest.mark.parametrize("world", [3, 4])
def test_more_ranks_than_blocks_per_chromosome_and_an_empty_rank(tmp_path, world):
    """Three LD blocks on three / four ranks (gloo on CPU): a rank may hold blocks of one chromosome only, or NO block at
    all -- it still takes part in every exchange (the partial sums of each EM iteration, the posterior all-gather at the
    end) and ends up with the posterior of all SNPs; the fit equals the single-process fit."""
    script = tmp_path / "worker.py"
    port = 29500 + ((os.getpid() * 11 + world * 197) % 2000)
    script.write_text(_WORKER_EMPTY.format(root=ROOT, port=port, world=world))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {r}" in o, o[-3000:]
