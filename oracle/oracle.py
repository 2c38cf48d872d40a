"""TEST INFRASTRUCTURE ONLY -- ctypes loaders for the parity oracle.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this module.  Nothing under ``viprs_amd/`` does.

Two back ends with the *same positional signatures as the reference's Cython boundary*
(``viprs/model/vi/e_step_cpp.pyx:91-195`` under /root/reference):

* ``restated``  -> ``oracle/liboracle.so``: the plain-C restatement (``estep_oracle.c``).
* ``reference`` -> ``oracle/_ref/libviprs_ref.so``: the reference's own ``e_step.hpp`` compiled from
  where it lies (``oracle/Makefile``); present whenever it was built in the authoring container.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_RESTATED = os.path.join(_HERE, "liboracle.so")
_LIB_REF = os.path.join(_HERE, "_ref", "libviprs_ref.so")
# the same sources with -march=x86-64-v3 (AVX2 + FMA): bench.py's "optimistic CPU" line only, never the parity reference
_LIB_REF_V3 = os.path.join(_HERE, "_ref", "libviprs_ref_v3.so")

_TCODE = {np.dtype(np.float32): 0, np.dtype(np.float64): 1}
_UCODE = {np.dtype(np.int8): 0, np.dtype(np.int16): 1, np.dtype(np.int32): 2,
          np.dtype(np.int64): 3, np.dtype(np.float32): 4, np.dtype(np.float64): 5}
_ICODE = {np.dtype(np.int32): 0, np.dtype(np.int64): 1}


def build(force=False):
    """Compile the restatement (and, if /root/reference is present, oracle/_ref)."""
    if force or not os.path.exists(_LIB_RESTATED) or not os.path.exists(_LIB_REF):
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)


def have_reference(kind="reference"):
    return os.path.exists(_LIB_REF_V3 if kind == "reference_v3" else _LIB_REF)


_libs = {}


def _lib(kind):
    if kind not in _libs:
        path = {"restated": _LIB_RESTATED, "reference": _LIB_REF, "reference_v3": _LIB_REF_V3}[kind]
        if not os.path.exists(path):
            build()
        _libs[kind] = ctypes.CDLL(path)
    return _libs[kind]


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _check_vec(name, a, dtype, ndim=1, order="C"):
    if not isinstance(a, np.ndarray) or a.dtype != dtype:
        raise ValueError(f"Buffer dtype mismatch for {name}: expected {dtype}, got {getattr(a, 'dtype', type(a))}")
    if a.ndim != ndim:
        raise ValueError(f"Buffer has wrong number of dimensions for {name} (expected {ndim}, got {a.ndim})")
    if order == "C" and not a.flags.c_contiguous:
        raise ValueError(f"ndarray {name} is not C-contiguous")
    if order == "F" and not a.flags.f_contiguous:
        raise ValueError(f"ndarray {name} is not Fortran contiguous")


def _common(ld_left_bound, ld_indptr, ld_data, std_beta):
    _check_vec("ld_left_bound", ld_left_bound, np.dtype(np.int32))
    if ld_indptr.dtype not in _ICODE:
        raise ValueError("Buffer dtype mismatch for ld_indptr")
    if ld_data.dtype not in _UCODE:
        raise ValueError("Buffer dtype mismatch for ld_data")
    if std_beta.dtype not in _TCODE:
        raise ValueError("Buffer dtype mismatch for std_beta")
    return _TCODE[std_beta.dtype], _UCODE[ld_data.dtype], _ICODE[ld_indptr.dtype]


def cpp_e_step(ld_left_bound, ld_indptr, ld_data, std_beta, var_gamma, var_mu, eta, q, eta_diff,
               u_logs, sqrt_half_var_tau, mu_mult, dq_scale, threads, low_memory, kind="restated"):
    """Same positional signature as e_step_cpp.pyx:91-122 (+ ``kind``)."""
    t, u, i = _common(ld_left_bound, ld_indptr, ld_data, std_beta)
    T = std_beta.dtype
    for n, a in (("var_gamma", var_gamma), ("var_mu", var_mu), ("eta", eta), ("q", q),
                 ("eta_diff", eta_diff), ("u_logs", u_logs), ("sqrt_half_var_tau", sqrt_half_var_tau),
                 ("mu_mult", mu_mult)):
        _check_vec(n, a, T)
    m = var_mu.shape[0]
    lib = _lib(kind)
    if kind != "restated":
        rc = lib.ref_e_step(t, u, i, ctypes.c_int(m), _p(ld_left_bound), _p(ld_indptr), _p(ld_data),
                            _p(std_beta), _p(var_gamma), _p(var_mu), _p(eta), _p(q), _p(eta_diff),
                            _p(u_logs), _p(sqrt_half_var_tau), _p(mu_mult), ctypes.c_double(dq_scale),
                            ctypes.c_int(threads), ctypes.c_int(bool(low_memory)))
    else:
        ip64 = np.ascontiguousarray(ld_indptr, dtype=np.int64)
        rc = lib.oracle_e_step(t, u, ctypes.c_int64(m), _p(ld_left_bound), _p(ip64), _p(ld_data),
                               _p(std_beta), _p(var_gamma), _p(var_mu), _p(eta), _p(q), _p(eta_diff),
                               _p(u_logs), _p(sqrt_half_var_tau), _p(mu_mult),
                               ctypes.c_double(dq_scale), ctypes.c_int(bool(low_memory)))
    if rc != 0:
        raise RuntimeError(f"oracle e_step failed with code {rc}")


def e_step_block_parallel(block_start, ld_left_bound, ld_indptr, ld_data, std_beta, var_gamma, var_mu, eta, q, eta_diff,
                          u_logs, sqrt_half_var_tau, mu_mult, dq_scale, n_threads, low_memory, kind="reference"):
    """The reference's e_step<T,U,I>(threads=1) once per LD block, blocks spread over `n_threads` OpenMP threads
    (largest first, dynamic schedule) inside oracle/_ref (ref_shim.cpp `ref_e_step_blocks`): the exact parallel
    CPU variant (bin/viprs_fit:1080-1086 at LD-block grain).  Same result as ONE threads=1 call, bit for bit."""
    t, u, i = _common(ld_left_bound, ld_indptr, ld_data, std_beta)
    T = std_beta.dtype
    for n, a in (("var_gamma", var_gamma), ("var_mu", var_mu), ("eta", eta), ("q", q),
                 ("eta_diff", eta_diff), ("u_logs", u_logs), ("sqrt_half_var_tau", sqrt_half_var_tau),
                 ("mu_mult", mu_mult)):
        _check_vec(n, a, T)
    bs = np.ascontiguousarray(block_start, dtype=np.int64)
    order = np.ascontiguousarray(np.argsort(-np.diff(bs), kind="stable"), dtype=np.int32)
    rc = _lib(kind).ref_e_step_blocks(t, u, i, ctypes.c_int(len(order)), _p(bs), _p(order), _p(ld_left_bound),
                                      _p(ld_indptr), _p(ld_data), _p(std_beta), _p(var_gamma), _p(var_mu), _p(eta),
                                      _p(q), _p(eta_diff), _p(u_logs), _p(sqrt_half_var_tau), _p(mu_mult),
                                      ctypes.c_double(dq_scale), ctypes.c_int(int(n_threads)),
                                      ctypes.c_int(bool(low_memory)))
    if rc != 0:
        raise RuntimeError(f"reference block-parallel e_step failed with code {rc}")


def cpp_e_step_mixture(ld_left_bound, ld_indptr, ld_data, std_beta, var_gamma, var_mu, eta, q,
                       eta_diff, log_null_pi, u_logs, sqrt_half_var_tau, mu_mult, dq_scale, threads,
                       low_memory, kind="restated"):
    """Same positional signature as e_step_cpp.pyx:125-159 (+ ``kind``)."""
    t, u, i = _common(ld_left_bound, ld_indptr, ld_data, std_beta)
    T = std_beta.dtype
    for n, a in (("eta", eta), ("q", q), ("eta_diff", eta_diff), ("log_null_pi", log_null_pi)):
        _check_vec(n, a, T)
    for n, a in (("var_gamma", var_gamma), ("var_mu", var_mu), ("u_logs", u_logs),
                 ("sqrt_half_var_tau", sqrt_half_var_tau), ("mu_mult", mu_mult)):
        _check_vec(n, a, T, ndim=2, order="C")
    m, K = var_mu.shape
    lib = _lib(kind)
    if kind != "restated":
        rc = lib.ref_e_step_mixture(t, u, i, ctypes.c_int(m), ctypes.c_int(K), _p(ld_left_bound),
                                    _p(ld_indptr), _p(ld_data), _p(std_beta), _p(var_gamma),
                                    _p(var_mu), _p(eta), _p(q), _p(eta_diff), _p(log_null_pi),
                                    _p(u_logs), _p(sqrt_half_var_tau), _p(mu_mult),
                                    ctypes.c_double(dq_scale), ctypes.c_int(threads),
                                    ctypes.c_int(bool(low_memory)))
    else:
        ip64 = np.ascontiguousarray(ld_indptr, dtype=np.int64)
        rc = lib.oracle_e_step_mixture(t, u, ctypes.c_int64(m), ctypes.c_int(K), _p(ld_left_bound),
                                       _p(ip64), _p(ld_data), _p(std_beta), _p(var_gamma), _p(var_mu),
                                       _p(eta), _p(q), _p(eta_diff), _p(log_null_pi), _p(u_logs),
                                       _p(sqrt_half_var_tau), _p(mu_mult), ctypes.c_double(dq_scale),
                                       ctypes.c_int(bool(low_memory)))
    if rc != 0:
        raise RuntimeError(f"oracle e_step_mixture failed with code {rc}")


def cpp_e_step_grid(ld_left_bound, ld_indptr, ld_data, std_beta, var_gamma, var_mu, eta, q, eta_diff,
                    u_logs, half_var_tau, mu_mult, dq_scale, active_model_idx, threads, low_memory,
                    kind="restated"):
    """Same positional signature as e_step_cpp.pyx:161-195 (+ ``kind``)."""
    t, u, i = _common(ld_left_bound, ld_indptr, ld_data, std_beta)
    T = std_beta.dtype
    for n, a in (("var_gamma", var_gamma), ("var_mu", var_mu), ("eta", eta), ("q", q),
                 ("eta_diff", eta_diff), ("u_logs", u_logs), ("half_var_tau", half_var_tau),
                 ("mu_mult", mu_mult)):
        _check_vec(n, a, T, ndim=2, order="F")
    if active_model_idx.dtype != np.int32:
        raise ValueError("Buffer dtype mismatch for active_model_idx")
    active = np.ascontiguousarray(active_model_idx)  # int[:] may be strided in the reference
    m = var_mu.shape[0]
    lib = _lib(kind)
    if kind != "restated":
        rc = lib.ref_e_step_grid(t, u, i, ctypes.c_int(m), ctypes.c_int(active.shape[0]), _p(active),
                                 _p(ld_left_bound), _p(ld_indptr), _p(ld_data), _p(std_beta),
                                 _p(var_gamma), _p(var_mu), _p(eta), _p(q), _p(eta_diff), _p(u_logs),
                                 _p(half_var_tau), _p(mu_mult), ctypes.c_double(dq_scale),
                                 ctypes.c_int(threads), ctypes.c_int(bool(low_memory)))
    else:
        ip64 = np.ascontiguousarray(ld_indptr, dtype=np.int64)
        rc = lib.oracle_e_step_grid(t, u, ctypes.c_int64(m), ctypes.c_int(active.shape[0]), _p(active),
                                    _p(ld_left_bound), _p(ip64), _p(ld_data), _p(std_beta),
                                    _p(var_gamma), _p(var_mu), _p(eta), _p(q), _p(eta_diff),
                                    _p(u_logs), _p(half_var_tau), _p(mu_mult),
                                    ctypes.c_double(dq_scale), ctypes.c_int(bool(low_memory)))
    if rc != 0:
        raise RuntimeError(f"oracle e_step_grid failed with code {rc}")


def check_blas_support():
    return bool(_lib("reference").ref_blas_supported())


def check_omp_support():
    return bool(_lib("reference").ref_omp_supported())


def expf_model(x):
    lib = _lib("restated")
    lib.oracle_expf_model.restype = ctypes.c_float
    return lib.oracle_expf_model(ctypes.c_float(x))


def expf_model_mismatches(lo_bits, hi_bits, stride=1):
    lib = _lib("restated")
    lib.oracle_expf_model_mismatches.restype = ctypes.c_int64
    return lib.oracle_expf_model_mismatches(ctypes.c_uint32(lo_bits), ctypes.c_uint32(hi_bits),
                                            ctypes.c_uint32(stride))


def exp_model(x):
    """The model of the host libm's double exp, x <= 0 (estep_oracle.c: what the device executes for a float64 state)."""
    lib = _lib("restated")
    lib.oracle_exp_model.restype = ctypes.c_double
    return lib.oracle_exp_model(ctypes.c_double(x))


def exp_model_mismatches(lo, step, n):
    """Bit mismatches of the model against exp() at x_i = -(lo + i step), i < n, and their two neighbours in the last place."""
    lib = _lib("restated")
    lib.oracle_exp_model_mismatches.restype = ctypes.c_int64
    return lib.oracle_exp_model_mismatches(ctypes.c_double(lo), ctypes.c_double(step), ctypes.c_int64(n))
