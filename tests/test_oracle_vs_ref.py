"""CPU: the plain-C restatement (oracle/estep_oracle.c) against the reference's own e_step.hpp
compiled from /root/reference (oracle/_ref), bit for bit, over every (T, U, I) combination of the
Cython boundary, both LD forms, several sweeps; plus the double-precision expf model that the
device executes, against the host libm."""
import struct

import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H
from viprs_amd.utils import synthetic as syn

needs_ref = pytest.mark.skipif(not O.have_reference(), reason="oracle/_ref not built (needs /root/reference)")


def _problem(sizes, low_memory, ld_dtype=np.float32, T=np.float32, indptr_dtype=np.int64, seed=21):
    return syn.make_problem(sizes=sizes, low_memory=low_memory, ld_dtype=ld_dtype, seed=seed,
                            indptr_dtype=indptr_dtype, float_precision=T)


@needs_ref
def test_reference_build_flags():
    # the reference built here has no CBLAS (deterministic fma loops) and has OpenMP
    assert O.check_blas_support() is False
    assert O.check_omp_support() is True


@needs_ref
@pytest.mark.parametrize("low_memory", [False, True])
@pytest.mark.parametrize("T", [np.float32, np.float64])
@pytest.mark.parametrize("ld_dtype", [np.int8, np.int16, np.int32, np.int64, np.float32, np.float64])
@pytest.mark.parametrize("indptr_dtype", [np.int32, np.int64])
def test_e_step_bit_exact_all_dtypes(low_memory, T, ld_dtype, indptr_dtype):
    ld, ss, inp = _problem([37, 128, 300], low_memory, ld_dtype, T, indptr_dtype)
    st0 = inp.state_copy()
    ref = H.run_oracle(ld, inp, st0, kind="reference", sweeps=3)
    got = H.run_oracle(ld, inp, st0, kind="restated", sweeps=3)
    H.assert_state_equal(got, ref)


@needs_ref
@pytest.mark.parametrize("low_memory", [False, True])
@pytest.mark.parametrize("ld_dtype", [np.int8, np.int16, np.float32])
@pytest.mark.parametrize("kind", ["longrange", "sample"])
def test_e_step_bit_exact_on_far_field_ld(low_memory, ld_dtype, kind):
    """LD whose far field matters (synthetic.py): the upper form's second pass (`dot`, e_step.hpp:82-104) now
    sums terms that are NOT negligible -- the reference build's summation order is the sequential fma chain."""
    ld, ss, inp = syn.make_problem(sizes=[37, 700, 300], low_memory=low_memory, ld_dtype=ld_dtype, seed=23, kind=kind)
    st0 = inp.state_copy()
    ref = H.run_oracle(ld, inp, st0, kind="reference", sweeps=3)
    got = H.run_oracle(ld, inp, st0, kind="restated", sweeps=3)
    H.assert_state_equal(got, ref)
    cut = H.run_oracle(H.cut_far_field(ld, 128), inp, st0, kind="reference", sweeps=1)
    assert (cut["q"] != H.run_oracle(ld, inp, st0, kind="reference", sweeps=1)["q"]).sum() > ld.m // 2


@needs_ref
def test_threads1_is_deterministic_and_skip_branch_is_hit():
    ld, ss, inp = _problem([500], False)
    st0 = inp.state_copy()
    a = H.run_oracle(ld, inp, st0, kind="reference")
    b = H.run_oracle(ld, inp, st0, kind="reference")
    H.assert_state_equal(a, b)
    skipped = a["eta_diff"] == 0
    assert skipped.sum() > 0                                     # first sweep skips some SNPs
    assert np.all(a["var_gamma"][skipped] == np.float32(inp.pi))  # ... and leaves them stale
    assert np.all(a["var_mu"][skipped] == 0)


def _mixture_inputs(ld, ss, K=4, T=np.float32, seed=5):
    m = ld.m
    rng = np.random.default_rng(seed)
    d = 2.0 ** np.linspace(-min(K - 1, 7), 0, K)                 # VIPRSMix.py:52
    pis = 0.01 * np.array([0.4, 0.3, 0.2, 0.1])[:K] if K == 4 else np.full(K, 0.01 / K)
    sigma_eps, h2 = 0.8, 0.2
    tau = d * (m * pis.sum() / h2)
    n = ss.n_per_snp[:, None]
    var_tau = n / sigma_eps + tau[None, :]
    mu_mult = (n / (var_tau * sigma_eps)).astype(T)
    u_logs = (np.log(pis) - np.log(1 - pis) + 0.5 * (np.log(tau) - np.log(var_tau))).astype(T)
    shvt = np.sqrt(0.5 * var_tau).astype(T)
    log_null_pi = np.full(m, np.log(1.0 - pis.sum()), dtype=T)
    st = dict(var_gamma=np.tile(pis.astype(T), (m, 1)), var_mu=(0.01 * rng.standard_normal((m, K))).astype(T),
              eta=np.zeros(m, T), q=np.zeros(m, T), eta_diff=np.zeros(m, T))
    return dict(log_null_pi=log_null_pi, u_logs=np.ascontiguousarray(u_logs), shvt=np.ascontiguousarray(shvt),
                mu_mult=np.ascontiguousarray(mu_mult)), st


@needs_ref
@pytest.mark.parametrize("low_memory", [False, True])
@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_mixture_bit_exact(low_memory, T):
    ld, ss, inp = _problem([50, 210], low_memory, T=T)
    mix, st0 = _mixture_inputs(ld, ss, 4, T)
    out = {}
    for kind in ("reference", "restated"):
        st = {k: v.copy() for k, v in st0.items()}
        for _ in range(3):
            O.cpp_e_step_mixture(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, inp.std_beta, st["var_gamma"],
                                 st["var_mu"], st["eta"], st["q"], st["eta_diff"], mix["log_null_pi"],
                                 mix["u_logs"], mix["shvt"], mix["mu_mult"], ld.dq_scale, 1, low_memory, kind=kind)
        out[kind] = st
    H.assert_state_equal(out["restated"], out["reference"])
    # closed-form invariant: eta == sum_k gamma_k mu_k
    s = out["reference"]
    np.testing.assert_allclose(s["eta"], (s["var_gamma"] * s["var_mu"]).sum(axis=1), rtol=1e-4, atol=1e-7)


def _grid_inputs(ld, ss, G=8, T=np.float32):
    m = ld.m
    pis = np.logspace(-3, -1, G)
    sig = np.linspace(0.7, 0.95, G)
    tau = pis * m / (1 - sig)
    n = ss.n_per_snp[:, None]
    var_tau = n / sig[None, :] + tau[None, :]
    mk = lambda a: np.asfortranarray(a.astype(T))
    return dict(u_logs=mk(np.log(pis) - np.log(1 - pis) + 0.5 * (np.log(tau) - np.log(var_tau))),
                hvt=mk(0.5 * var_tau), mu_mult=mk(n / (var_tau * sig[None, :]))), \
        dict(var_gamma=mk(np.tile(pis, (m, 1))), var_mu=mk(np.zeros((m, G))), eta=mk(np.zeros((m, G))),
             q=mk(np.zeros((m, G))), eta_diff=mk(np.zeros((m, G))))


@needs_ref
@pytest.mark.parametrize("low_memory", [False, True])
def test_grid_bit_exact_and_active_subset(low_memory):
    ld, ss, inp = _problem([64, 150], low_memory)
    g, st0 = _grid_inputs(ld, ss, 8)
    active = np.array([5, 1, 6], dtype=np.int32)
    out = {}
    for kind in ("reference", "restated"):
        st = {k: v.copy(order="F") for k, v in st0.items()}
        for _ in range(2):
            O.cpp_e_step_grid(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, inp.std_beta, st["var_gamma"], st["var_mu"],
                              st["eta"], st["q"], st["eta_diff"], g["u_logs"], g["hvt"], g["mu_mult"], ld.dq_scale,
                              active, 1, low_memory, kind=kind)
        out[kind] = st
    H.assert_state_equal(out["restated"], out["reference"])
    untouched = [c for c in range(8) if c not in active]
    assert np.all(out["reference"]["eta"][:, untouched] == 0)


def test_q_is_R_minus_I_times_eta():
    # closed-form invariant of the sweep from q = eta = 0: q == (R - I) eta
    ld, ss, inp = _problem([120], False)
    st = H.run_oracle(ld, inp, inp.state_copy(), kind="restated", sweeps=4)
    b = 120
    R = ld.ld_data.reshape(b, b).astype(np.float64)
    q = (R - np.eye(b)) @ st["eta"].astype(np.float64)
    np.testing.assert_allclose(st["q"], q, rtol=2e-4, atol=2e-6)


def test_expf_model_matches_host_libm_on_a_sweep():
    """The double-precision expf model the device executes (viprs_amd/csrc/device_math.h, mirrored
    in oracle/estep_oracle.c) against this host's expf: dense windows + a strided sweep of every
    float in [-104, 0] (the full 1.1e9-value sweep takes ~25 s of 8 cores; 0 mismatches)."""
    lo = struct.unpack("<I", struct.pack("<f", -0.0))[0]
    hi = struct.unpack("<I", struct.pack("<f", -104.0))[0]
    assert O.expf_model_mismatches(lo, hi, stride=251) == 0
    one = struct.unpack("<I", struct.pack("<f", -1.0))[0]
    assert O.expf_model_mismatches(one, one + 2_000_000, 1) == 0
    big = struct.unpack("<I", struct.pack("<f", -80.0))[0]
    assert O.expf_model_mismatches(big, big + 2_000_000, 1) == 0
    # -63.09946...: the one input in range where glibc's non-FMA build of the same code differs
    # from the FMA ifunc variant this host runs; the model follows the FMA variant
    x = float.fromhex("-0x1.f8cbb2p+5")
    xb = struct.unpack("<I", struct.pack("<f", x))[0]
    assert O.expf_model_mismatches(xb, xb, 1) == 0


def test_exp_model_matches_host_libm():
    """The model of glibc's DOUBLE exp the device executes for a float64 state (viprs_amd/csrc/device_math.h:
    exp_glibc_f64_*, mirrored in oracle/estep_oracle.c; constants read from this host's libm by tools/extract_glibc_exp.py)
    against this host's exp(): dense and coarse sweeps of x <= 0 with both neighbours in the last place of every point --
    the normal range, the subnormal results of x in (-745.2, -708.4), the special path (-1024, -512], |x| < 2^-54, and
    beyond -1024."""
    assert O.exp_model_mismatches(0.0, 1.000001 / (1 << 16), 3_000_000) == 0           # [0, 45.8) dense
    assert O.exp_model_mismatches(1e-3, 760.0 / 2_000_000, 2_000_000) == 0            # [0, 760) coarse
    assert O.exp_model_mismatches(511.9, 0.0003, 2_000_000) == 0                      # the special-case path
    assert O.exp_model_mismatches(0.0, 2.0 ** -60, 100_000) == 0 and O.exp_model_mismatches(2.0 ** -54 - 2.0 ** -70, 2.0 ** -80, 100_000) == 0
    assert O.exp_model_mismatches(1023.9, 0.001, 1000) == 0
    import math
    for x in (0.0, -0.0, float("-inf"), -745.13321910194122, -745.2, -708.3964185322641, -1e308, -512.0, -1024.0):
        assert struct.pack("<d", math.exp(x)) == struct.pack("<d", O.exp_model(x)), x
    assert math.isnan(O.exp_model(float("nan")))


def test_generated_exp_tables_are_in_sync():
    """oracle/exp_glibc_f64_tab.h (the checker's copy) and viprs_amd/csrc/exp_glibc_f64_tab.h (the device's) are the same
    generated file (tools/extract_glibc_exp.py)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a = open(os.path.join(root, "oracle", "exp_glibc_f64_tab.h")).read()
    b = open(os.path.join(root, "viprs_amd", "csrc", "exp_glibc_f64_tab.h")).read()
    assert a == b and "VIPRS_EXP64_TAB_INIT" in a
