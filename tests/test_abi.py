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


def test_shipped_library_carries_no_experiment_switch():
    """kernels_common.h: timing / profiling switches (some produce wrong results) are recorded per translation unit;
    the library the tests load -- the one that ships -- must report none."""
    assert L.build_flags() == "", f"libviprs_hip.so was built with experiment switches: {L.build_flags()!r}"


def test_experiment_switch_without_the_gate_does_not_compile():
    """-DPANEL_TIMING_NO_SECOND_PASS (wrong results) without -DVIPRS_EXPERIMENTAL must be a compile error; every
    `#ifdef` knob of the kernel headers must be on the guarded list."""
    import shutil
    import subprocess
    import pytest
    csrc = os.path.join(ROOT, "viprs_amd", "csrc")
    common = open(os.path.join(csrc, "kernels_common.h")).read()
    knobs = set()
    for fn in os.listdir(csrc):
        if fn.endswith((".h", ".inc", ".hip", ".cpp")):
            for mm in re.finditer(r"^\s*#\s*(?:ifdef|ifndef|if\s+defined\(?|elif\s+defined\(?)\s*([A-Z][A-Z0-9_]+)",
                                  open(os.path.join(csrc, fn)).read(), flags=re.M):
                knobs.add(mm.group(1))
    knobs -= {"VIPRS_EXPERIMENTAL", "VIPRS_HIP_H"}
    knobs = {k for k in knobs if not k.startswith("VIPRS_BF_")}
    assert knobs, "no knobs found: the scan is broken"
    for k in sorted(knobs):
        assert f"defined({k})" in common, f"{k} is an #ifdef switch of the kernels but not on kernels_common.h's guarded list"
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    cmd = [hipcc, "-std=c++17", "--offload-arch=gfx950", "-fsyntax-only", "-x", "hip", "-DPANEL_TIMING_NO_SECOND_PASS",
           os.path.join(csrc, "kernels_common.h")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode != 0 and "VIPRS_EXPERIMENTAL" in r.stderr
    r = subprocess.run(cmd + ["-DVIPRS_EXPERIMENTAL"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-500:]


def test_plan_cache_fingerprint_sees_in_place_edits():
    """The drop-in entry points keep LD resident by buffer identity; the per-call content fingerprint must change when the
    caller edits the arrays in place (small arrays: every byte; large ones: both ends + 256 sampled windows + length)."""
    import numpy as np
    from viprs_amd.vi import e_step_hip as S
    small = np.arange(1000, dtype=np.float32)
    f0 = S._fingerprint(small)
    small[517] += 1
    assert S._fingerprint(small) != f0
    big = np.zeros(1 << 22, dtype=np.int8)
    f0 = S._fingerprint(big)
    for pos in (0, 4095, big.size - 1, big.size - 4096, ((big.size - 64) // S._FP_SAMPLES) * 100 + 5):
        big[pos] = 1
        assert S._fingerprint(big) != f0, pos
        big[pos] = 0
    assert S._fingerprint(big) == f0
    assert S._fingerprint(big[:-1]) != f0                      # the length is part of it
    lb = np.zeros(4, np.int32)
    assert S._fingerprint(lb, small) != S._fingerprint(small, lb)
