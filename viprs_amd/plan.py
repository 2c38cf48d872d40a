"""Python handles over the C ABI: ``LDPlan`` (device-resident LD + block schedule) and
``DeviceState`` (device-resident variational state).

``LDPlan`` replaces the "load LD matrices to memory" step of ``VIPRS.__init__``
(viprs/model/VIPRS.py:151-172 in the reference): the LD arrays are validated, partitioned into
independent LD blocks, uploaded and re-laid-out for the panel kernels once; every later E-step
only moves per-SNP vectors.
"""
import ctypes
import os
import weakref

import numpy as np

from . import _lib as L

_FLOAT_CODE = {np.dtype(np.float32): L.F32, np.dtype(np.float64): L.F64}
_LD_CODE = {np.dtype(np.int8): L.LD_I8, np.dtype(np.int16): L.LD_I16, np.dtype(np.int32): L.LD_I32,
            np.dtype(np.int64): L.LD_I64, np.dtype(np.float32): L.LD_F32, np.dtype(np.float64): L.LD_F64}
_IP_CODE = {np.dtype(np.int32): L.IP_I32, np.dtype(np.int64): L.IP_I64}


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _check_index_arrays(ld_left_bound, ld_indptr):
    # same errors the Cython memoryviews raise (e_step_cpp.pyx:91-93: int[::1], indptr_type[::1])
    if not isinstance(ld_left_bound, np.ndarray) or ld_left_bound.dtype != np.int32:
        raise ValueError("Buffer dtype mismatch, expected 'int' but got "
                         f"'{getattr(ld_left_bound, 'dtype', type(ld_left_bound))}' (ld_left_bound)")
    if not isinstance(ld_indptr, np.ndarray) or ld_indptr.dtype not in _IP_CODE:
        raise ValueError("Buffer dtype mismatch, expected int32/int64 ld_indptr but got "
                         f"'{getattr(ld_indptr, 'dtype', type(ld_indptr))}'")
    for name, a in (("ld_left_bound", ld_left_bound), ("ld_indptr", ld_indptr)):
        if a.ndim != 1:
            raise ValueError(f"Buffer has wrong number of dimensions (expected 1, got {a.ndim}) ({name})")
        if not a.flags.c_contiguous:
            raise ValueError(f"ndarray is not C-contiguous ({name})")
    if ld_indptr.shape[0] != ld_left_bound.shape[0] + 1:
        raise ValueError("ld_indptr must have len(ld_left_bound) + 1 entries")


def plan_blocks(ld_left_bound, ld_indptr, low_memory):
    """Host-only block discovery (no GPU needed).  Returns ``(block_start, block_kind)`` with
    ``block_start`` of length ``n_blocks + 1``."""
    _check_index_arrays(ld_left_bound, ld_indptr)
    m = ld_left_bound.shape[0]
    n = ctypes.c_int64(0)
    starts = np.zeros(m + 1, dtype=np.int64)
    kinds = np.zeros(max(m, 1), dtype=np.int32)
    L.check(L.lib.viprs_plan_blocks(m, _ptr(ld_left_bound), _ptr(ld_indptr), _IP_CODE[ld_indptr.dtype],
                                    int(bool(low_memory)), ctypes.byref(n), _ptr(starts), _ptr(kinds)))
    return starts[: n.value + 1].copy(), kinds[: n.value].copy()


class LDPlan:
    """Device-resident LD matrix of one chromosome (or any set of LD blocks)."""

    def __init__(self, ld_left_bound, ld_indptr, ld_data, low_memory, device=0, math_mode="exact"):
        _check_index_arrays(ld_left_bound, ld_indptr)
        if not isinstance(ld_data, np.ndarray) or ld_data.dtype not in _LD_CODE:
            raise ValueError("Buffer dtype mismatch for ld_data: "
                             f"'{getattr(ld_data, 'dtype', type(ld_data))}' is not a supported LD dtype")
        if ld_data.ndim != 1 or not ld_data.flags.c_contiguous:
            raise ValueError("ld_data must be a C-contiguous 1-d array")
        self.m = int(ld_left_bound.shape[0])
        if self.m and int(ld_indptr[-1]) != ld_data.shape[0]:
            raise ValueError("ld_indptr[-1] must equal len(ld_data)")
        self.low_memory = bool(low_memory)
        self.ld_dtype = ld_data.dtype
        self.device = int(device)
        self._h = ctypes.c_void_p()
        L.check(L.lib.viprs_plan_create(ctypes.byref(self._h), self.m, _ptr(ld_left_bound), _ptr(ld_indptr),
                                        _IP_CODE[ld_indptr.dtype], _ptr(ld_data), _LD_CODE[ld_data.dtype],
                                        int(self.low_memory), self.device))
        self.set_math_mode(math_mode)

    @classmethod
    def from_upper(cls, ld_indptr, ld_data, diag_value=None, device=0, math_mode="exact"):
        """Symmetric plan (``low_memory=False`` arithmetic) from the compact upper-triangular store
        (row j = correlations with SNPs j+1 .. j+len_j): uploaded once and mirrored into the symmetric
        windows on the device -- the host never builds ``ld_mat.load(return_symmetric=True)``
        (VIPRS.py:167-172).  ``diag_value`` defaults to 1 for float LD and to the quantisation maximum
        (``np.iinfo(dtype).max``) for integer LD."""
        if not isinstance(ld_indptr, np.ndarray) or ld_indptr.dtype not in _IP_CODE:
            raise ValueError("Buffer dtype mismatch, expected int32/int64 ld_indptr but got "
                             f"'{getattr(ld_indptr, 'dtype', type(ld_indptr))}'")
        if ld_indptr.ndim != 1 or not ld_indptr.flags.c_contiguous or ld_indptr.shape[0] < 1:
            raise ValueError("ld_indptr must be a C-contiguous 1-d array with m + 1 entries")
        if not isinstance(ld_data, np.ndarray) or ld_data.dtype not in _LD_CODE:
            raise ValueError("Buffer dtype mismatch for ld_data: "
                             f"'{getattr(ld_data, 'dtype', type(ld_data))}' is not a supported LD dtype")
        if ld_data.ndim != 1 or not ld_data.flags.c_contiguous:
            raise ValueError("ld_data must be a C-contiguous 1-d array")
        self = cls.__new__(cls)
        self.m = int(ld_indptr.shape[0]) - 1
        if self.m and int(ld_indptr[-1]) != ld_data.shape[0]:
            raise ValueError("ld_indptr[-1] must equal len(ld_data)")
        if diag_value is None:
            diag_value = float(np.iinfo(ld_data.dtype).max) if np.issubdtype(ld_data.dtype, np.integer) else 1.0
        self.low_memory = False
        self.ld_dtype = ld_data.dtype
        self.device = int(device)
        self._h = ctypes.c_void_p()
        L.check(L.lib.viprs_plan_create_expanded(ctypes.byref(self._h), self.m, _ptr(ld_indptr),
                                                 _IP_CODE[ld_indptr.dtype], _ptr(ld_data), _LD_CODE[ld_data.dtype],
                                                 float(diag_value), self.device))
        self.set_math_mode(math_mode)
        return self

    @classmethod
    def synthetic(cls, ld, device=0, math_mode="exact"):
        """Plan of a `viprs_amd.utils.synthetic` "longrange" workload whose LD entries are GENERATED ON THE DEVICE
        (`viprs_plan_create_synthetic`): `ld` is the `SyntheticLD` skeleton (`make_ld(..., kind="longrange",
        data=False)`; a full one works too, its `ld_data` is ignored).  Bit-identical to `LDPlan(ld.ld_left_bound,
        ld.ld_indptr, make_ld(...).ld_data, ...)` (tests/test_synth_device.py); measurement support, not on the
        reference's path."""
        from .utils import synthetic as syn
        if ld.kind != "longrange" or ld.params is None:
            raise ValueError("LDPlan.synthetic: a 'longrange' SyntheticLD with its block parameters is needed")
        dt = np.dtype(ld.ld_dtype if ld.ld_data is None else ld.ld_data.dtype)
        if dt not in (np.dtype(np.float32), np.dtype(np.int8), np.dtype(np.int16)):
            raise ValueError("LDPlan.synthetic: float32, int8 or int16 LD")
        sizes = np.ascontiguousarray(np.diff(ld.block_start), dtype=np.int64)
        vecs = syn.longrange_device_params(sizes, ld.rho, ld.params)
        self = cls.__new__(cls)
        self.m = int(sizes.sum())
        self.low_memory = bool(ld.low_memory)
        self.ld_dtype = dt
        self.device = int(device)
        self._h = ctypes.c_void_p()
        L.check(L.lib.viprs_plan_create_synthetic(ctypes.byref(self._h), int(sizes.shape[0]), _ptr(sizes),
                                                  *[_ptr(v) for v in vecs], _LD_CODE[dt], int(self.low_memory),
                                                  self.device))
        self.set_math_mode(math_mode)
        return self

    def windows(self):
        """``(ld_left_bound int32, ld_indptr int64)`` of the plan's rows."""
        lb = np.zeros(self.m, dtype=np.int32)
        ip = np.zeros(self.m + 1, dtype=np.int64)
        L.check(L.lib.viprs_plan_get_windows(self.handle, _ptr(lb), _ptr(ip)))
        return lb, ip

    # -- lifetime -----------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            # a DeviceState dereferences its plan when it is destroyed (closing the plan first would be a
            # use-after-free on the C side): the plan closes the states that are still open before it goes
            for st in list(getattr(self, "_states", ())):
                st.close()
            L.lib.viprs_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        if not self._h:
            raise ValueError("LDPlan is closed")
        return self._h

    # -- queries ------------------------------------------------------------------------------
    def info(self, key):
        v = ctypes.c_int64(0)
        L.check(L.lib.viprs_plan_info(self.handle, key, ctypes.byref(v)))
        return v.value

    @property
    def n_blocks(self):
        return self.info(L.INFO_N_BLOCKS)

    @property
    def nnz(self):
        return self.info(L.INFO_NNZ)

    def blocks(self):
        n = self.n_blocks
        starts = np.zeros(n + 1, dtype=np.int64)
        kinds = np.zeros(max(n, 1), dtype=np.int32)
        L.check(L.lib.viprs_plan_get_blocks(self.handle, _ptr(starts), _ptr(kinds)))
        return starts, kinds[:n]

    def set_active_blocks(self, active):
        """Which LD blocks the following sweeps visit: a boolean per block in SNP order (`blocks()`), None = all of them
        (`viprs_plan_set_active_blocks`).  Spike-and-slab / mixture states; the batched grid kernel refuses a filtered plan."""
        if active is None:
            L.check(L.lib.viprs_plan_set_active_blocks(self.handle, None, 0))
            return
        a = np.ascontiguousarray(active, dtype=np.uint8)
        L.check(L.lib.viprs_plan_set_active_blocks(self.handle, _ptr(a), int(a.shape[0])))

    def set_math_mode(self, mode):
        code = {"exact": L.MATH_EXACT, "fast": L.MATH_FAST}.get(mode, mode)
        L.check(L.lib.viprs_plan_set_math_mode(self.handle, int(code)))
        self.math_mode = "exact" if code == L.MATH_EXACT else "fast"

    def last_kernel_ms(self, which=0):
        ms = ctypes.c_double(0.0)
        L.check(L.lib.viprs_plan_last_kernel_ms(self.handle, int(which), ctypes.byref(ms)))
        return ms.value

    def timing_reset(self):
        L.check(L.lib.viprs_plan_timing_reset(self.handle))

    def timing_history(self, which=0, capacity=256):
        """HIP-event durations (ms) of the most recent sweeps, oldest first (which: 0 = whole sweep,
        1 = panel kernel(s) only)."""
        buf = (ctypes.c_double * capacity)()
        n = ctypes.c_int(0)
        L.check(L.lib.viprs_plan_timing_history(self.handle, int(which), buf, capacity, ctypes.byref(n)))
        return [buf[i] for i in range(n.value)]

    def effective_math_mode(self):
        """'exact' / 'fast' / 'mixed': what the kernels of the last sweep really computed in (fast has no instantiation for
        mixtures of 9+ components or float64 states: those run exact whatever `set_math_mode` was given); None before
        the first sweep."""
        mask = ctypes.c_int(0)
        L.check(L.lib.viprs_plan_last_math_modes(self.handle, ctypes.byref(mask)))
        return {0: None, 1: "exact", 2: "fast", 3: "mixed"}[mask.value & 3]

    def last_skipped(self):
        n = ctypes.c_int64(0)
        L.check(L.lib.viprs_plan_last_skipped(self.handle, ctypes.byref(n)))
        return n.value

    # -- one-shot host-buffer E-steps (drop-ins for the Cython entry points) -------------------
    def e_step(self, std_beta, var_gamma, var_mu, eta, q, eta_diff, u_logs, sqrt_half_var_tau, mu_mult,
               dq_scale, threads=1, low_memory=None):
        T = _FLOAT_CODE[std_beta.dtype]
        low_memory = self.low_memory if low_memory is None else low_memory
        L.check(L.lib.viprs_e_step(self.handle, T, _ptr(std_beta), _ptr(var_gamma), _ptr(var_mu), _ptr(eta),
                                   _ptr(q), _ptr(eta_diff), _ptr(u_logs), _ptr(sqrt_half_var_tau),
                                   _ptr(mu_mult), float(dq_scale), int(threads), int(bool(low_memory))))

    def e_step_mixture(self, std_beta, var_gamma, var_mu, eta, q, eta_diff, log_null_pi, u_logs,
                       sqrt_half_var_tau, mu_mult, dq_scale, threads=1, low_memory=None):
        T = _FLOAT_CODE[std_beta.dtype]
        low_memory = self.low_memory if low_memory is None else low_memory
        K = var_mu.shape[1]
        L.check(L.lib.viprs_e_step_mixture(self.handle, T, int(K), _ptr(std_beta), _ptr(var_gamma), _ptr(var_mu),
                                           _ptr(eta), _ptr(q), _ptr(eta_diff), _ptr(log_null_pi), _ptr(u_logs),
                                           _ptr(sqrt_half_var_tau), _ptr(mu_mult), float(dq_scale),
                                           int(threads), int(bool(low_memory))))

    def e_step_grid(self, std_beta, var_gamma, var_mu, eta, q, eta_diff, u_logs, half_var_tau, mu_mult,
                    dq_scale, active_model_idx, threads=1, low_memory=None):
        T = _FLOAT_CODE[std_beta.dtype]
        low_memory = self.low_memory if low_memory is None else low_memory
        G = var_mu.shape[1]
        active = np.ascontiguousarray(active_model_idx, dtype=np.int32)
        L.check(L.lib.viprs_e_step_grid(self.handle, T, int(G), _ptr(std_beta), _ptr(var_gamma), _ptr(var_mu),
                                        _ptr(eta), _ptr(q), _ptr(eta_diff), _ptr(u_logs), _ptr(half_var_tau),
                                        _ptr(mu_mult), float(dq_scale), _ptr(active), int(active.shape[0]),
                                        int(threads), int(bool(low_memory))))


class DeviceState:
    """Variational state + per-SNP inputs of one plan, resident in HBM across EM iterations."""

    FIELDS = {
        "std_beta": L.FIELD_STD_BETA, "u_logs": L.FIELD_U_LOGS,
        "sqrt_half_var_tau": L.FIELD_SQRT_HALF_VAR_TAU, "half_var_tau": L.FIELD_SQRT_HALF_VAR_TAU,
        "mu_mult": L.FIELD_MU_MULT, "log_null_pi": L.FIELD_LOG_NULL_PI, "var_gamma": L.FIELD_VAR_GAMMA,
        "var_mu": L.FIELD_VAR_MU, "eta": L.FIELD_ETA, "q": L.FIELD_Q, "eta_diff": L.FIELD_ETA_DIFF,
    }

    # Placement probe (see `_probe_placement`): candidates tried / smallest plan it is worth it for
    PLACEMENT_CANDIDATES = 6
    PLACEMENT_MIN_SNPS = 200_000

    def __init__(self, plan, float_precision="float32", model="spike_slab", width=1, placement=None):
        self.plan = plan
        self.dtype = np.dtype(float_precision)
        self.model = model
        self.width = int(width)
        self._kind = {"spike_slab": L.MODEL_SPIKE_SLAB, "mixture": L.MODEL_MIXTURE, "grid": L.MODEL_GRID}[model]
        self._h = self._create()
        self.placement = None
        if not hasattr(plan, "_states"):
            plan._states = weakref.WeakSet()
        plan._states.add(self)
        mode = placement if placement is not None else os.environ.get("VIPRS_STATE_PLACEMENT", "probe")
        # (stream-bound sweeps only: fp32 state on the panel kernels -- dense blocks, mixtures of up to 8 components; a plan of
        #  windowed components or a wide mixture is chain-bound and has no panel-kernel bracket to rank candidates by)
        if (mode == "probe" and self.dtype == np.float32 and model in ("spike_slab", "mixture") and self.width <= 8
                and plan.m >= self.PLACEMENT_MIN_SNPS and plan.info(L.INFO_N_DENSE) > 0):
            own = self._h
            try:
                self._probe_placement()
            except Exception as e:              # noqa: BLE001 -- a usable state exists: the probe must never fail the constructor
                # (out of memory on the extra candidates, a team hand-off time-out when another process shares the GPU ...)
                if self._h is not own and self._h:
                    try:
                        L.lib.viprs_state_destroy(self._h)
                    except Exception:           # noqa: BLE001
                        pass
                self._h = own
                self.placement = {"error": f"{type(e).__name__}: {e}"[:200], "chosen": 0}
                try:                            # hand out what the caller was promised: a zeroed state
                    self._zero_fields()
                    self.plan.timing_reset()
                except Exception:               # noqa: BLE001
                    pass

    def _create(self):
        h = ctypes.c_void_p()
        L.check(L.lib.viprs_state_create(ctypes.byref(h), self.plan.handle, _FLOAT_CODE[self.dtype], self._kind, self.width))
        return h

    def _probe_placement(self):
        """WHERE the allocator puts a state's per-SNP arrays moves the stream-bound sweeps by up to 8 % (four levels
        2.6 % apart on cfg3: same plan, same kernels, same memory counters -- EXPERIMENTS.md round 5); the level is fixed
        for the life of the allocation.  So a large fp32 state is allocated `PLACEMENT_CANDIDATES` times, each candidate
        sweeps a synthetic input a few times (a sweep from the standard start on typical hyper-parameters: every LD
        entry is streamed), the fastest one is kept and handed out zeroed, the others are freed.  ~0.1 s once per fit;
        `VIPRS_STATE_PLACEMENT=off` (or `placement="off"`) skips it.  `self.placement` records what was measured."""
        from .utils import synthetic as syn
        m, K, T = self.plan.m, self.width, self.dtype
        rng = np.random.default_rng(12345)

        class _SS:
            n_per_snp = np.full(m, 1e5)
        beta = (0.01 * rng.standard_normal(m)).astype(T)
        if self.model == "spike_slab":
            inp = syn.make_inputs(type("S", (), {"std_beta": beta, "n_per_snp": _SS.n_per_snp})())
            fields = {"std_beta": beta, "u_logs": inp.u_logs, "sqrt_half_var_tau": inp.sqrt_half_var_tau, "mu_mult": inp.mu_mult}
            pi0 = inp.pi
        else:
            mix = syn.make_mixture_inputs(_SS, K, float_precision=T)
            pi0 = mix.pop("pi")
            fields = dict(mix, std_beta=beta)
        itemsize = np.dtype(self.plan.ld_dtype).itemsize
        dq = 1.0 if np.issubdtype(np.dtype(self.plan.ld_dtype), np.floating) else 1.0 / (2 ** (8 * itemsize - 1) - 1)
        own, handles, times = self._h, [self._h], []
        try:
            for _ in range(self.PLACEMENT_CANDIDATES - 1):
                handles.append(self._create())

            def sweeps(h, n):
                self._h = h
                self.plan.timing_reset()
                for _ in range(n):
                    self.reset(pi0)
                    self.e_step(dq, sync=False)
                self.synchronize()
                return self.plan.timing_history(which=1)
            for h in handles:
                self._h = h
                for k, a in fields.items():
                    self.upload(k, a)
            sweeps(handles[0], 40)                                   # clocks up before anything is compared
            times = [[] for _ in handles]
            for _ in range(2):                                       # two rounds: no candidate is only measured early
                for i, h in enumerate(handles):
                    times[i] += list(sweeps(h, 5))
            mins = [min(t) if t else 0.0 for t in times]
            # no decision when the timings say nothing: no panel-kernel bracket (zeros) or all candidates level (< 0.2 %)
            decided = min(mins) > 0.0 and (max(mins) - min(mins)) > 2e-3 * min(mins)
            best = int(np.argmin(mins)) if decided else 0
            self._h = handles[best]
            self._zero_fields()                                      # handed out as a fresh state: all zeros
            self.plan.timing_reset()
            self.placement = {"candidates": len(handles), "chosen": best, "decided": bool(decided),
                              "kernel_ms_min": [round(float(x), 4) for x in mins]}
        finally:
            keep = self._h if self._h in handles else own
            for h in handles:
                if h is not keep and h:
                    L.lib.viprs_state_destroy(h)
            self._h = keep

    def _zero_fields(self):
        T, m = self.dtype, self.plan.m
        zero = {k: np.zeros(self._shape(k), dtype=T) for k in
                ("std_beta", "u_logs", "sqrt_half_var_tau", "mu_mult", "var_gamma", "var_mu", "eta", "q", "eta_diff")}
        if self.model == "mixture":
            zero["log_null_pi"] = np.zeros(m, dtype=T)
        for k, a in zero.items():
            self.upload(k, a)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            L.lib.viprs_state_destroy(self._h)
            self._h = ctypes.c_void_p()
            self.plan._states.discard(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _shape(self, name):
        m = self.plan.m
        if self.model == "spike_slab" or name in ("std_beta", "log_null_pi"):
            return (m,)
        if self.model == "mixture":
            return (m,) if name in ("eta", "q", "eta_diff") else (m, self.width)
        return (m, self.width)

    def upload(self, name, array):
        shape = self._shape(name)
        order = "F" if self.model == "grid" and len(shape) == 2 else "C"
        if array.dtype != self.dtype:
            raise ValueError(f"Buffer dtype mismatch for {name}: expected {self.dtype}, got {array.dtype}")
        if tuple(array.shape) != shape:
            raise ValueError(f"{name}: expected shape {shape}, got {array.shape}")
        a = np.asarray(array, order=order)
        if (order == "F" and not a.flags.f_contiguous) or (order == "C" and not a.flags.c_contiguous):
            raise ValueError(f"ndarray is not {'Fortran' if order == 'F' else 'C'} contiguous ({name})")
        L.check(L.lib.viprs_state_upload(self._h, self.FIELDS[name], _ptr(a)))

    def download(self, name, out=None):
        shape = self._shape(name)
        order = "F" if self.model == "grid" and len(shape) == 2 else "C"
        if out is None:
            out = np.empty(shape, dtype=self.dtype, order=order)
        else:                                   # the C side copies field_elems * itemsize bytes whatever it is handed
            if not isinstance(out, np.ndarray) or out.dtype != self.dtype:
                raise ValueError(f"Buffer dtype mismatch for {name}: expected {self.dtype}, got "
                                 f"{getattr(out, 'dtype', type(out))}")
            if tuple(out.shape) != shape:
                raise ValueError(f"{name}: expected shape {shape}, got {out.shape}")
            if (order == "F" and not out.flags.f_contiguous) or (order == "C" and not out.flags.c_contiguous):
                raise ValueError(f"ndarray is not {'Fortran' if order == 'F' else 'C'} contiguous ({name})")
            if not out.flags.writeable:
                raise ValueError(f"{name}: output array is read-only")
        L.check(L.lib.viprs_state_download(self._h, self.FIELDS[name], _ptr(out)))
        return out

    def reset(self, pi):
        L.check(L.lib.viprs_state_reset(self._h, float(pi)))

    def set_comm(self, comm):
        """Attach an ``RcclComm`` (None detaches): `sums_begin / sums_end` (and the mixture / grid-column
        variants) then return the sums over ALL ranks -- one all-gather + rank-ordered reduction on the
        plan's stream per call (`viprs_state_set_comm`)."""
        self._comm = comm                       # keeps the communicator alive as long as the state uses it
        L.check(L.lib.viprs_state_set_comm(self._h, comm.handle if comm is not None else None))

    # -- device-resident EM iteration (spike-and-slab) ------------------------------------------
    def set_n_per_snp(self, n_per_snp):
        n = np.ascontiguousarray(n_per_snp, dtype=np.float64)
        if n.shape != (self.plan.m,):
            raise ValueError(f"n_per_snp: expected shape ({self.plan.m},), got {n.shape}")
        L.check(L.lib.viprs_state_set_n_per_snp(self._h, _ptr(n)))

    def set_snp_weights(self, weights):
        """Per-SNP weights of sum [0] (None clears): 1 / (SNPs of the chromosome) when several chromosomes
        share this plan, so that sum [0] is the reference's sum of per-chromosome means of gamma."""
        if weights is None:
            L.check(L.lib.viprs_state_set_snp_weights(self._h, None))
            return
        w = np.ascontiguousarray(weights, dtype=np.float64)
        if w.shape != (self.plan.m,):
            raise ValueError(f"weights: expected shape ({self.plan.m},), got {w.shape}")
        L.check(L.lib.viprs_state_set_snp_weights(self._h, _ptr(w)))

    def prep(self, logit_pi, log_tau_beta, sigma_epsilon, tau_beta, one_plus_lambda):
        """VIPRS.py:400-418 on the device (asynchronous on the plan's stream).  The scalars are
        evaluated by the caller (reference dtype semantics)."""
        L.check(L.lib.viprs_state_prep(self._h, float(logit_pi), float(log_tau_beta), float(sigma_epsilon),
                                       float(tau_beta), float(one_plus_lambda)))

    def sums(self, one_plus_lambda):
        """The M-step / ELBO partial sums of this plan's SNPs (float64, deterministic order)."""
        out = (ctypes.c_double * L.N_SUMS)()
        L.check(L.lib.viprs_state_sums(self._h, float(one_plus_lambda), out))
        return np.array(out[:], dtype=np.float64)

    def sums_begin(self, one_plus_lambda):
        """Enqueue the reduction (asynchronous); `sums_end` collects it.  Lets several chromosomes reduce
        concurrently instead of one synchronisation per chromosome."""
        L.check(L.lib.viprs_state_sums_begin(self._h, float(one_plus_lambda)))

    def sums_end(self):
        out = (ctypes.c_double * L.N_SUMS)()
        L.check(L.lib.viprs_state_sums_end(self._h, out))
        return np.array(out[:], dtype=np.float64)

    # -- SNP groups: one model per chromosome in one state ---------------------------------------
    def set_groups(self, group_start):
        """Contiguous SNP ranges (whole LD blocks) with their own hyper-parameters and sums: `group_start` has
        n_groups + 1 entries from 0 to m; None removes the groups."""
        if group_start is None:
            L.check(L.lib.viprs_state_set_groups(self._h, 0, None))
            self.n_groups = 0
            return
        gs = np.ascontiguousarray(group_start, dtype=np.int64)
        L.check(L.lib.viprs_state_set_groups(self._h, int(gs.shape[0]) - 1, _ptr(gs)))
        self.n_groups = int(gs.shape[0]) - 1

    def prep_groups(self, params):
        """`prep` with per-group scalars: rows (group, logit_pi, log_tau_beta, sigma_epsilon, tau_beta, one_plus_lambda)."""
        p = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, 6)
        L.check(L.lib.viprs_state_prep_groups(self._h, int(p.shape[0]), _ptr(p)))

    def sums_groups_begin(self, groups, one_plus_lambda):
        r = np.ascontiguousarray(np.column_stack([np.asarray(groups, dtype=np.float64),
                                                  np.broadcast_to(np.asarray(one_plus_lambda, dtype=np.float64),
                                                                  (len(groups),))]))
        self._n_sum_cols = int(r.shape[0])
        L.check(L.lib.viprs_state_sums_groups_begin(self._h, self._n_sum_cols, _ptr(r)))

    def sums_groups_end(self):
        out = np.zeros((self._n_sum_cols, L.N_SUMS), dtype=np.float64)
        L.check(L.lib.viprs_state_sums_groups_end(self._h, _ptr(out)))
        return out

    def prep_mixture_groups(self, params):
        """`prep_mixture` with per-group parameters: rows (group, log_null_pi, sigma_epsilon, one_plus_lambda, logit_pi[K],
        log_tau_beta[K], tau_beta[K])."""
        p = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, 4 + 3 * self.width)
        L.check(L.lib.viprs_state_prep_mixture_groups(self._h, int(p.shape[0]), _ptr(p)))

    def sums_mixture_groups_begin(self, groups, one_plus_lambda):
        r = np.ascontiguousarray(np.column_stack([np.asarray(groups, dtype=np.float64),
                                                  np.broadcast_to(np.asarray(one_plus_lambda, dtype=np.float64),
                                                                  (len(groups),))]))
        self._n_sum_cols = int(r.shape[0])
        L.check(L.lib.viprs_state_sums_mixture_groups_begin(self._h, self._n_sum_cols, _ptr(r)))

    def sums_mixture_groups_end(self):
        """(n, 7 + 6 K) rows in the layout of `sums_mixture_end`."""
        out = np.zeros((self._n_sum_cols, 7 + 6 * self.width), dtype=np.float64)
        L.check(L.lib.viprs_state_sums_mixture_groups_end(self._h, _ptr(out)))
        return out

    # -- one model (column) of a grid state -----------------------------------------------------
    def prep_column(self, g, logit_pi, log_tau_beta, sigma_epsilon, tau_beta, one_plus_lambda):
        L.check(L.lib.viprs_state_prep_column(self._h, int(g), float(logit_pi), float(log_tau_beta),
                                              float(sigma_epsilon), float(tau_beta), float(one_plus_lambda)))

    def sums_column(self, g, one_plus_lambda):
        out = (ctypes.c_double * L.N_SUMS)()
        L.check(L.lib.viprs_state_sums_column(self._h, int(g), float(one_plus_lambda), out))
        return np.array(out[:], dtype=np.float64)

    # -- device-resident EM iteration of a mixture state (K <= 8) --------------------------------------
    def set_log_var_tau(self, log_var_tau):
        a = np.ascontiguousarray(log_var_tau, dtype=np.float64)
        if a.shape != (self.plan.m, self.width):
            raise ValueError(f"log_var_tau: expected shape ({self.plan.m}, {self.width}), got {a.shape}")
        L.check(L.lib.viprs_state_set_log_var_tau(self._h, _ptr(a)))

    def prep_mixture(self, logit_pi, log_tau_beta, tau_beta, log_null_pi, sigma_epsilon, one_plus_lambda):
        v = [np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=np.float64), (self.width,))) for x in
             (logit_pi, log_tau_beta, tau_beta)]
        L.check(L.lib.viprs_state_prep_mixture(self._h, _ptr(v[0]), _ptr(v[1]), _ptr(v[2]), float(log_null_pi),
                                               float(sigma_epsilon), float(one_plus_lambda)))

    def sums_mixture_begin(self, one_plus_lambda):
        L.check(L.lib.viprs_state_sums_mixture_begin(self._h, float(one_plus_lambda)))

    def sums_mixture_end(self):
        out = np.zeros(7 + 6 * self.width, dtype=np.float64)
        L.check(L.lib.viprs_state_sums_mixture_end(self._h, _ptr(out)))
        return out

    def prep_columns(self, params):
        """`prep_column` for several models in one launch: rows (column, logit_pi, log_tau_beta,
        sigma_epsilon, tau_beta, one_plus_lambda)."""
        p = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, 6)
        L.check(L.lib.viprs_state_prep_columns(self._h, int(p.shape[0]), _ptr(p)))

    def sums_columns_begin(self, cols, one_plus_lambda):
        c = np.ascontiguousarray(np.column_stack([np.asarray(cols, dtype=np.float64),
                                                  np.broadcast_to(np.asarray(one_plus_lambda, dtype=np.float64),
                                                                  (len(cols),))]))
        self._n_sum_cols = int(c.shape[0])
        L.check(L.lib.viprs_state_sums_columns_begin(self._h, self._n_sum_cols, _ptr(c)))

    def sums_columns_end(self):
        out = np.zeros((self._n_sum_cols, L.N_SUMS), dtype=np.float64)
        L.check(L.lib.viprs_state_sums_columns_end(self._h, _ptr(out)))
        return out

    def reset_column(self, g, pi):
        L.check(L.lib.viprs_state_reset_column(self._h, int(g), float(pi)))

    def e_step(self, dq_scale=1.0, active_model_idx=None, sync=True):
        if active_model_idx is not None:
            active = np.ascontiguousarray(active_model_idx, dtype=np.int32)
            L.check(L.lib.viprs_state_e_step(self._h, float(dq_scale), _ptr(active), int(active.shape[0]),
                                             int(bool(sync))))
        else:
            L.check(L.lib.viprs_state_e_step(self._h, float(dq_scale), None, 0, int(bool(sync))))

    def synchronize(self):
        L.check(L.lib.viprs_state_synchronize(self._h))
