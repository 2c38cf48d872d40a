"""Drop-in replacements for the reference's Cython boundary ``viprs.model.vi.e_step_cpp``
(viprs/model/vi/e_step_cpp.pyx:71-195): the same function names, positional signatures, in-place
update contract and dtype/layout errors, executed by the HIP kernels of ``libviprs_hip.so``.

The LD arrays are uploaded to the GPU the first time a given ``(ld_left_bound, ld_indptr,
ld_data, low_memory)`` combination is seen and stay resident (``plan_for``); later calls move
only the per-SNP vectors (and re-upload when a content fingerprint of the arrays has changed).  ``threads`` is accepted and ignored: results always follow the
reference's ``threads=1`` semantics.
"""
from collections import OrderedDict

import numpy as np

from .. import _lib as L
from ..plan import LDPlan

_PLAN_CACHE = OrderedDict()
# Device-resident LD kept by the drop-in entry points: bounded by BYTES of device memory (a genome-scale plan is
# ~4 GB), least recently used plans are closed first.  VIPRS_PLAN_CACHE_GB overrides the default budget.
import os as _os
_PLAN_CACHE_BYTES = int(float(_os.environ.get("VIPRS_PLAN_CACHE_GB", "64")) * (1 << 30))
_DEFAULT_DEVICE = 0
_DEFAULT_MATH = "exact"


def set_default_device(device):
    global _DEFAULT_DEVICE
    _DEFAULT_DEVICE = int(device)


def set_default_math_mode(mode):
    """'exact' (bit-for-bit the reference's arithmetic) or 'fast' (hardware exp/rcp sigmoid)."""
    global _DEFAULT_MATH
    if mode not in ("exact", "fast"):
        raise ValueError("math mode must be 'exact' or 'fast'")
    _DEFAULT_MATH = mode
    for plan, _, _, _ in _PLAN_CACHE.values():
        plan.set_math_mode(mode)


def clear_plan_cache():
    for plan, _, _, _ in _PLAN_CACHE.values():
        plan.close()
    _PLAN_CACHE.clear()


def set_plan_cache_budget(n_bytes):
    """Device-memory budget (bytes) of the plan cache; evicts down to it at once."""
    global _PLAN_CACHE_BYTES
    _PLAN_CACHE_BYTES = int(n_bytes)
    _evict(keep=None)


def _evict(keep):
    total = sum(b for _, _, b, _ in _PLAN_CACHE.values())
    for key in list(_PLAN_CACHE):
        if total <= _PLAN_CACHE_BYTES:
            break
        if key == keep:
            continue
        plan, _, nbytes, _ = _PLAN_CACHE.pop(key)
        plan.close()
        total -= nbytes


def invalidate(ld_data=None):
    """Forget the device copy of `ld_data` (all cached plans when None).  The cache is keyed on the IDENTITY of the
    host buffers; `plan_for` also fingerprints a sample of their contents on every call (`_fingerprint`), which catches
    an in-place edit unless it misses every sampled byte: call this after one to be certain."""
    if ld_data is None:
        clear_plan_cache()
        return
    addr = ld_data.__array_interface__["data"][0]
    for key in [k for k in _PLAN_CACHE if k[3] == addr]:
        plan, _, _, _ = _PLAN_CACHE.pop(key)
        plan.close()


_FP_EDGE = 4096          # bytes hashed at both ends of every array
_FP_SAMPLES = 256        # + this many 64-byte windows spread evenly over the array (viprs_host_fingerprint, abi_plan.hip)


def _fingerprint(*arrays):
    """Cheap content fingerprint of the LD arrays (a few microseconds per call, whatever the array size): per array its
    length, the first and last 4 KB and 256 evenly spaced 64-byte windows (`viprs_host_fingerprint`, host code of the
    library).  The reference's Cython function reads the caller's memory on every call (e_step_cpp.pyx:91-122); the
    resident device copy can only notice edits this way."""
    import ctypes
    out = ctypes.c_uint64(0)
    h = 0
    for a in arrays:
        if not a.flags.c_contiguous:
            a = np.ascontiguousarray(a)
        L.check(L.lib.viprs_host_fingerprint(a.ctypes.data_as(ctypes.c_void_p), a.nbytes, ctypes.byref(out)))
        h = (h * 0x100000001b3 + out.value) & 0xFFFFFFFFFFFFFFFF
    return h


def plan_for(ld_left_bound, ld_indptr, ld_data, low_memory):
    """The cached device plan for these LD arrays.  The key is the identity of the buffers (the arrays are kept alive
    by the cache so the identity cannot be recycled); a content fingerprint taken on every call replaces the plan when
    the caller has edited the arrays in place since the upload."""
    key = (ld_left_bound.__array_interface__["data"][0], ld_left_bound.shape[0],
           ld_indptr.__array_interface__["data"][0], ld_data.__array_interface__["data"][0],
           ld_data.shape[0], str(ld_data.dtype), bool(low_memory), _DEFAULT_DEVICE)
    fp = _fingerprint(ld_left_bound, ld_indptr, ld_data)
    hit = _PLAN_CACHE.get(key)
    if hit is not None:
        if hit[3] == fp:
            _PLAN_CACHE.move_to_end(key)
            return hit[0]
        _PLAN_CACHE.pop(key)[0].close()           # edited in place since the upload: the device copy is stale
    plan = LDPlan(ld_left_bound, ld_indptr, ld_data, low_memory, device=_DEFAULT_DEVICE, math_mode=_DEFAULT_MATH)
    _PLAN_CACHE[key] = (plan, (ld_left_bound, ld_indptr, ld_data), int(plan.info(L.INFO_LD_BYTES_DEVICE)), fp)
    _evict(keep=key)
    return plan


def check_blas_support():
    """e_step_cpp.pyx:71-72.  No BLAS on the device path."""
    return bool(L.lib.viprs_check_blas_support())


def check_omp_support():
    """e_step_cpp.pyx:75-76.  No OpenMP on the device path."""
    return bool(L.lib.viprs_check_omp_support())


def _floating(name, a, dtype, ndim=1, order="C"):
    # the errors Cython's typed memoryviews raise for floating[::1] / [:, ::1] / [::1, :]
    if not isinstance(a, np.ndarray):
        raise TypeError(f"{name}: expected a numpy array")
    if a.dtype != dtype:
        raise ValueError(f"Buffer dtype mismatch, expected '{dtype}' but got '{a.dtype}' ({name})")
    if a.ndim != ndim:
        raise ValueError(f"Buffer has wrong number of dimensions (expected {ndim}, got {a.ndim}) ({name})")
    if order == "C" and not a.flags.c_contiguous:
        raise ValueError(f"ndarray is not C-contiguous ({name})")
    if order == "F" and not a.flags.f_contiguous:
        raise ValueError(f"ndarray is not Fortran contiguous ({name})")


def _float_dtype(std_beta):
    if not isinstance(std_beta, np.ndarray) or std_beta.dtype not in (np.float32, np.float64):
        raise ValueError("Buffer dtype mismatch, expected float32/float64 (std_beta)")
    return std_beta.dtype


def cpp_e_step(ld_left_bound, ld_indptr, ld_data, std_beta, var_gamma, var_mu, eta, q, eta_diff, u_logs,
               sqrt_half_var_tau, mu_mult, dq_scale, threads, low_memory):
    """e_step_cpp.pyx:91-122 -> e_step.hpp:343-442 (spike-and-slab E-step), on the GPU."""
    T = _float_dtype(std_beta)
    m = var_mu.shape[0] if isinstance(var_mu, np.ndarray) else -1
    for name, a in (("std_beta", std_beta), ("var_gamma", var_gamma), ("var_mu", var_mu), ("eta", eta), ("q", q),
                    ("eta_diff", eta_diff), ("u_logs", u_logs), ("sqrt_half_var_tau", sqrt_half_var_tau),
                    ("mu_mult", mu_mult)):
        _floating(name, a, T)
        if a.shape[0] != m:
            raise ValueError(f"{name}: expected {m} entries, got {a.shape[0]}")
    plan = plan_for(ld_left_bound, ld_indptr, ld_data, low_memory)
    if plan.m != m:
        raise ValueError(f"LD arrays describe {plan.m} SNPs but the state vectors have {m}")
    plan.e_step(std_beta, var_gamma, var_mu, eta, q, eta_diff, u_logs, sqrt_half_var_tau, mu_mult, dq_scale,
                threads, low_memory)


def cpp_e_step_mixture(ld_left_bound, ld_indptr, ld_data, std_beta, var_gamma, var_mu, eta, q, eta_diff,
                       log_null_pi, u_logs, sqrt_half_var_tau, mu_mult, dq_scale, threads, low_memory):
    """e_step_cpp.pyx:125-159 -> e_step.hpp:447-551 (sparse-mixture E-step), on the GPU."""
    T = _float_dtype(std_beta)
    for name, a in (("std_beta", std_beta), ("eta", eta), ("q", q), ("eta_diff", eta_diff),
                    ("log_null_pi", log_null_pi)):
        _floating(name, a, T)
    for name, a in (("var_gamma", var_gamma), ("var_mu", var_mu), ("u_logs", u_logs),
                    ("sqrt_half_var_tau", sqrt_half_var_tau), ("mu_mult", mu_mult)):
        _floating(name, a, T, ndim=2, order="C")
    m, K = var_mu.shape
    for name, a in (("std_beta", std_beta), ("eta", eta), ("q", q), ("eta_diff", eta_diff), ("log_null_pi", log_null_pi)):
        if a.shape != (m,):
            raise ValueError(f"{name}: expected shape ({m},), got {a.shape}")
    for name, a in (("var_gamma", var_gamma), ("u_logs", u_logs), ("sqrt_half_var_tau", sqrt_half_var_tau),
                    ("mu_mult", mu_mult)):
        if a.shape != (m, K):
            raise ValueError(f"{name}: expected shape ({m}, {K}), got {a.shape}")
    plan = plan_for(ld_left_bound, ld_indptr, ld_data, low_memory)
    if plan.m != var_mu.shape[0]:
        raise ValueError(f"LD arrays describe {plan.m} SNPs but the state has {var_mu.shape[0]}")
    plan.e_step_mixture(std_beta, var_gamma, var_mu, eta, q, eta_diff, log_null_pi, u_logs, sqrt_half_var_tau,
                        mu_mult, dq_scale, threads, low_memory)


def cpp_e_step_grid(ld_left_bound, ld_indptr, ld_data, std_beta, var_gamma, var_mu, eta, q, eta_diff, u_logs,
                    half_var_tau, mu_mult, dq_scale, active_model_idx, threads, low_memory):
    """e_step_cpp.pyx:161-195 -> e_step.hpp:555-647 (grid of spike-and-slab models), on the GPU."""
    T = _float_dtype(std_beta)
    _floating("std_beta", std_beta, T)
    for name, a in (("var_gamma", var_gamma), ("var_mu", var_mu), ("eta", eta), ("q", q), ("eta_diff", eta_diff),
                    ("u_logs", u_logs), ("half_var_tau", half_var_tau), ("mu_mult", mu_mult)):
        _floating(name, a, T, ndim=2, order="F")
    if not isinstance(active_model_idx, np.ndarray) or active_model_idx.dtype != np.int32:
        raise ValueError("Buffer dtype mismatch, expected 'int' (active_model_idx)")
    m, G = var_mu.shape
    if std_beta.shape != (m,):
        raise ValueError(f"std_beta: expected shape ({m},), got {std_beta.shape}")
    for name, a in (("var_gamma", var_gamma), ("eta", eta), ("q", q), ("eta_diff", eta_diff), ("u_logs", u_logs),
                    ("half_var_tau", half_var_tau), ("mu_mult", mu_mult)):
        if a.shape != (m, G):
            raise ValueError(f"{name}: expected shape ({m}, {G}), got {a.shape}")
    if active_model_idx.ndim != 1 or (active_model_idx.size and (active_model_idx.min() < 0 or active_model_idx.max() >= G)):
        raise ValueError("active_model_idx: model index out of range")
    plan = plan_for(ld_left_bound, ld_indptr, ld_data, low_memory)
    if plan.m != var_mu.shape[0]:
        raise ValueError(f"LD arrays describe {plan.m} SNPs but the state has {var_mu.shape[0]}")
    plan.e_step_grid(std_beta, var_gamma, var_mu, eta, q, eta_diff, u_logs, half_var_tau, mu_mult, dq_scale,
                     active_model_idx, threads, low_memory)
