"""ctypes binding of ``libviprs_hip.so`` (C ABI declared in ``include/viprs_hip.h``).

The product path has no CPU fallback: if the HIP library is missing or fails to load, importing
this module raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C viprs_amd/csrc``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VIPRS_HIP_LIB: development override (e.g. an instrumented build); the default is the in-tree library
LIB_PATH = os.environ.get("VIPRS_HIP_LIB") or os.path.join(_HERE, "lib", "libviprs_hip.so")

# dtype / enum codes (include/viprs_hip.h)
F32, F64 = 0, 1
LD_I8, LD_I16, LD_I32, LD_I64, LD_F32, LD_F64 = range(6)
IP_I32, IP_I64 = 0, 1
MATH_EXACT, MATH_FAST = 0, 1
BLOCK_DENSE_SYM, BLOCK_DENSE_UPPER, BLOCK_RAGGED = 0, 1, 2
MODEL_SPIKE_SLAB, MODEL_MIXTURE, MODEL_GRID = 0, 1, 2
(FIELD_STD_BETA, FIELD_U_LOGS, FIELD_SQRT_HALF_VAR_TAU, FIELD_MU_MULT, FIELD_LOG_NULL_PI,
 FIELD_VAR_GAMMA, FIELD_VAR_MU, FIELD_ETA, FIELD_Q, FIELD_ETA_DIFF) = range(10)
(INFO_M, INFO_NNZ, INFO_N_BLOCKS, INFO_N_DENSE, INFO_N_RAGGED, INFO_MAX_BLOCK, INFO_LD_BYTES_DEVICE,
 INFO_LD_ELEM_SIZE, INFO_DEVICE, INFO_LOW_MEMORY, INFO_N_CU) = range(11)

OK, EINVAL, ELAYOUT, EDEVICE, ENOMEM, EUNSUPPORTED = 0, -1, -2, -3, -4, -5
N_SUMS = 11
COMM_ID_BYTES = 128


class ViprsHipError(RuntimeError):
    """A libviprs_hip call failed (device error, unsupported configuration...)."""


class ViprsLayoutError(ValueError):
    """The LD index arrays violate the contiguous-window contract."""


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: the MI355X E-step has no CPU fallback. "
        "Build it with `make -C viprs_amd/csrc` (hipcc --offload-arch=gfx950).")

lib = ctypes.CDLL(LIB_PATH)

_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_d = ctypes.c_double
_pi64 = ctypes.POINTER(ctypes.c_int64)
_pi32 = ctypes.POINTER(ctypes.c_int32)

_PROTOS = {
    "viprs_last_error": (ctypes.c_char_p, []),
    "viprs_version": (ctypes.c_char_p, []),
    "viprs_build_flags": (ctypes.c_char_p, []),
    "viprs_device_count": (_i, [ctypes.POINTER(_i)]),
    "viprs_host_fingerprint": (_i, [_vp, _i64, ctypes.POINTER(ctypes.c_uint64)]),
    "viprs_check_blas_support": (_i, []),
    "viprs_check_omp_support": (_i, []),
    "viprs_plan_blocks": (_i, [_i64, _vp, _vp, _i, _i, _pi64, _vp, _vp]),
    "viprs_plan_create": (_i, [ctypes.POINTER(_vp), _i64, _vp, _vp, _i, _vp, _i, _i, _i]),
    "viprs_plan_create_expanded": (_i, [ctypes.POINTER(_vp), _i64, _vp, _i, _vp, _i, ctypes.c_double, _i]),
    "viprs_plan_get_windows": (_i, [_vp, _vp, _vp]),
    "viprs_plan_destroy": (_i, [_vp]),
    "viprs_plan_info": (_i, [_vp, _i, _pi64]),
    "viprs_plan_get_blocks": (_i, [_vp, _vp, _vp]),
    "viprs_plan_set_math_mode": (_i, [_vp, _i]),
    "viprs_plan_set_active_blocks": (_i, [_vp, _vp, _i64]),
    "viprs_e_step": (_i, [_vp, _i] + [_vp] * 9 + [_d, _i, _i]),
    "viprs_e_step_mixture": (_i, [_vp, _i, _i] + [_vp] * 10 + [_d, _i, _i]),
    "viprs_e_step_grid": (_i, [_vp, _i, _i] + [_vp] * 9 + [_d, _vp, _i, _i, _i]),
    "viprs_state_create": (_i, [ctypes.POINTER(_vp), _vp, _i, _i, _i]),
    "viprs_state_destroy": (_i, [_vp]),
    "viprs_state_upload": (_i, [_vp, _i, _vp]),
    "viprs_state_download": (_i, [_vp, _i, _vp]),
    "viprs_state_reset": (_i, [_vp, _d]),
    "viprs_state_e_step": (_i, [_vp, _d, _vp, _i, _i]),
    "viprs_state_synchronize": (_i, [_vp]),
    "viprs_state_set_n_per_snp": (_i, [_vp, _vp]),
    "viprs_state_prep": (_i, [_vp, _d, _d, _d, _d, _d]),
    "viprs_state_set_snp_weights": (_i, [_vp, _vp]),
    "viprs_state_sums": (_i, [_vp, _d, ctypes.POINTER(_d)]),
    "viprs_state_sums_begin": (_i, [_vp, _d]),
    "viprs_state_sums_end": (_i, [_vp, ctypes.POINTER(_d)]),
    "viprs_state_prep_column": (_i, [_vp, _i, _d, _d, _d, _d, _d]),
    "viprs_state_sums_column": (_i, [_vp, _i, _d, ctypes.POINTER(_d)]),
    "viprs_state_set_log_var_tau": (_i, [_vp, _vp]),
    "viprs_state_prep_mixture": (_i, [_vp, _vp, _vp, _vp, _d, _d, _d]),
    "viprs_state_sums_mixture_begin": (_i, [_vp, _d]),
    "viprs_state_sums_mixture_end": (_i, [_vp, _vp]),
    "viprs_state_prep_columns": (_i, [_vp, _i, _vp]),
    "viprs_state_sums_columns_begin": (_i, [_vp, _i, _vp]),
    "viprs_state_sums_columns_end": (_i, [_vp, _vp]),
    "viprs_state_reset_column": (_i, [_vp, _i, _d]),
    "viprs_state_set_groups": (_i, [_vp, _i, _vp]),
    "viprs_state_prep_groups": (_i, [_vp, _i, _vp]),
    "viprs_state_sums_groups_begin": (_i, [_vp, _i, _vp]),
    "viprs_state_sums_groups_end": (_i, [_vp, _vp]),
    "viprs_state_prep_mixture_groups": (_i, [_vp, _i, _vp]),
    "viprs_state_sums_mixture_groups_begin": (_i, [_vp, _i, _vp]),
    "viprs_state_sums_mixture_groups_end": (_i, [_vp, _vp]),
    "viprs_comm_unique_id": (_i, [_vp]),
    "viprs_comm_create": (_i, [ctypes.POINTER(_vp), _vp, _i, _i, _i]),
    "viprs_comm_destroy": (_i, [_vp]),
    "viprs_comm_rank": (_i, [_vp, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "viprs_comm_allreduce": (_i, [_vp, _vp, _i, _i]),
    "viprs_comm_allgather": (_i, [_vp, _vp, _i64, _vp]),
    "viprs_comm_barrier": (_i, [_vp]),
    "viprs_state_set_comm": (_i, [_vp, _vp]),
    "viprs_device_synchronize": (_i, [_i]),
    "viprs_plan_create_synthetic": (_i, [ctypes.POINTER(_vp), _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i]),
    "viprs_synthetic_ld_host": (_i, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i64]),
    "viprs_plan_last_kernel_ms": (_i, [_vp, _i, ctypes.POINTER(_d)]),
    "viprs_plan_last_skipped": (_i, [_vp, _pi64]),
    "viprs_plan_last_math_modes": (_i, [_vp, ctypes.POINTER(_i)]),
    "viprs_plan_timing_reset": (_i, [_vp]),
    "viprs_plan_timing_history": (_i, [_vp, _i, ctypes.POINTER(_d), _i, ctypes.POINTER(_i)]),
}

EXPORTED_SYMBOLS = tuple(_PROTOS)

for _name, (_res, _args) in _PROTOS.items():
    _fn = getattr(lib, _name)        # AttributeError here = header/library mismatch
    _fn.restype = _res
    _fn.argtypes = _args


def last_error():
    msg = lib.viprs_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc):
    """Raise the Python exception matching a negative status code."""
    if rc == OK:
        return
    msg = last_error()
    if rc == ELAYOUT:
        raise ViprsLayoutError(msg)
    if rc == EINVAL:
        raise ValueError(msg)
    if rc == ENOMEM:
        raise MemoryError(msg)
    if rc == EUNSUPPORTED:
        raise NotImplementedError(msg)
    raise ViprsHipError(msg)


def build_flags():
    """Experiment switches the loaded library was compiled with ('' for the shipped build; include/viprs_hip.h)."""
    return (lib.viprs_build_flags() or b"").decode()


def device_count():
    n = _i(0)
    rc = lib.viprs_device_count(ctypes.byref(n))
    return n.value if rc == OK else 0
