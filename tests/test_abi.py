"""CPU: the C-ABI library loads without a GPU and exports every symbol include/viprs_hip.h declares."""
import ctypes
import os
import re

from viprs_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "viprs_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(viprs_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    declared = _declared_functions()
    assert len(declared) >= 20
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in viprs_hip.h but not exported"
        assert name in L.EXPORTED_SYMBOLS, f"{name} has no ctypes prototype in viprs_amd/_lib.py"
    for name in L.EXPORTED_SYMBOLS:
        assert name in declared, f"{name} bound in _lib.py but missing from viprs_hip.h"


def test_host_only_entry_points_work_without_gpu():
    assert L.lib.viprs_version().decode().startswith("viprs_amd")
    assert L.lib.viprs_check_blas_support() == 0 and L.lib.viprs_check_omp_support() == 0
    assert isinstance(L.device_count(), int)


def test_shim_has_the_reference_signatures():
    import inspect
    from viprs_amd.vi import e_step_hip as S
    assert list(inspect.signature(S.cpp_e_step).parameters) == [
        "ld_left_bound", "ld_indptr", "ld_data", "std_beta", "var_gamma", "var_mu", "eta", "q", "eta_diff",
        "u_logs", "sqrt_half_var_tau", "mu_mult", "dq_scale", "threads", "low_memory"]
    assert list(inspect.signature(S.cpp_e_step_mixture).parameters) == [
        "ld_left_bound", "ld_indptr", "ld_data", "std_beta", "var_gamma", "var_mu", "eta", "q", "eta_diff",
        "log_null_pi", "u_logs", "sqrt_half_var_tau", "mu_mult", "dq_scale", "threads", "low_memory"]
    assert list(inspect.signature(S.cpp_e_step_grid).parameters) == [
        "ld_left_bound", "ld_indptr", "ld_data", "std_beta", "var_gamma", "var_mu", "eta", "q", "eta_diff",
        "u_logs", "half_var_tau", "mu_mult", "dq_scale", "active_model_idx", "threads", "low_memory"]
    assert S.check_blas_support() is False and S.check_omp_support() is False


def test_shim_dtype_and_layout_errors_match_the_cython_boundary():
    import numpy as np
    import pytest
    from viprs_amd.vi import e_step_hip as S
    m = 4
    lb = np.zeros(m, np.int32); ip = np.arange(0, 4 * m + 1, m).astype(np.int64); ld = np.ones(4 * m, np.float32)
    f = lambda: np.zeros(m, np.float32)
    with pytest.raises(ValueError, match="dtype mismatch"):       # mixed f32/f64
        S.cpp_e_step(lb, ip, ld, f(), f().astype(np.float64), f(), f(), f(), f(), f(), f(), f(), 1.0, 1, False)
    with pytest.raises(ValueError, match="not C-contiguous"):
        S.cpp_e_step(lb, ip, ld, np.zeros(2 * m, np.float32)[::2], f(), f(), f(), f(), f(), f(), f(), f(), 1.0, 1, False)
    F = lambda: np.zeros((m, 3), np.float32, order="C")
    with pytest.raises(ValueError, match="Fortran"):              # grid wants column-major
        S.cpp_e_step_grid(lb, ip, ld, f(), F(), F(), F(), F(), F(), F(), F(), F(), 1.0, np.arange(3, dtype=np.int32), 1, False)


def test_shim_rejects_mismatched_shapes_before_touching_the_device():
    """The C side copies m * K elements whatever it is handed: the shims check every array against var_mu's shape."""
    import numpy as np
    import pytest
    from viprs_amd.vi import e_step_hip as S
    m, K = 6, 3
    lb = np.zeros(m, np.int32); ip = np.arange(0, m * m + 1, m).astype(np.int64); ld = np.ones(m * m, np.float32)
    v = lambda n=m: np.zeros(n, np.float32)
    C = lambda r=m, c=K: np.zeros((r, c), np.float32)
    with pytest.raises(ValueError, match="u_logs"):
        S.cpp_e_step_mixture(lb, ip, ld, v(), C(), C(), v(), v(), v(), v(), C(m, K + 1), C(), C(), 1.0, 1, False)
    with pytest.raises(ValueError, match="log_null_pi"):
        S.cpp_e_step_mixture(lb, ip, ld, v(), C(), C(), v(), v(), v(), v(m - 1), C(), C(), C(), 1.0, 1, False)
    F = lambda r=m, c=K: np.zeros((r, c), np.float32, order="F")
    with pytest.raises(ValueError, match="eta"):
        S.cpp_e_step_grid(lb, ip, ld, v(), F(), F(), F(m, K - 1), F(), F(), F(), F(), F(), 1.0, np.arange(K, dtype=np.int32), 1, False)
    with pytest.raises(ValueError, match="out of range"):
        S.cpp_e_step_grid(lb, ip, ld, v(), F(), F(), F(), F(), F(), F(), F(), F(), 1.0, np.array([0, K], dtype=np.int32), 1, False)


def test_plan_cache_is_bounded_by_bytes_and_can_be_invalidated():
    from viprs_amd.vi import e_step_hip as S
    assert S._PLAN_CACHE_BYTES >= 1 << 30 and callable(S.invalidate) and callable(S.set_plan_cache_budget)
