"""The device-side synthetic-LD generator (csrc/synth.hip; bench.py's per-rank workloads) against `synthetic.make_ld`:
the SAME entry function runs on the host (`viprs_synthetic_ld_host`, checked here bit for bit on the CPU) and in the
kernel (GPU test: a generated plan and an uploaded plan give `==` states, and the oracle agrees)."""
import ctypes

import numpy as np
import pytest

from tests import helpers as H
from viprs_amd import _lib as L
from viprs_amd.plan import _LD_CODE, _ptr
from viprs_amd.utils import synthetic as syn


def _host_generate(ld):
    sizes = np.ascontiguousarray(np.diff(ld.block_start), dtype=np.int64)
    vecs = syn.longrange_device_params(sizes, ld.rho, ld.params)
    dt = np.dtype(ld.ld_dtype)
    out = np.full(ld.nnz + 3, 77, dtype=dt)
    L.check(L.lib.viprs_synthetic_ld_host(int(sizes.shape[0]), _ptr(sizes), *[_ptr(v) for v in vecs], _LD_CODE[dt],
                                          int(ld.low_memory), _ptr(out), ld.nnz))
    assert (out[ld.nnz:] == 77).all()
    return out[:ld.nnz]


@pytest.mark.parametrize("low_memory", [False, True], ids=["sym", "upper"])
@pytest.mark.parametrize("ld_dtype", [np.float32, np.int8, np.int16], ids=["f32", "i8", "i16"])
def test_host_generator_equals_make_ld(low_memory, ld_dtype):
    sizes = [1, 2, 63, 64, 65, 300, 517, 5]
    ld = syn.make_ld(sizes, low_memory=low_memory, ld_dtype=ld_dtype, seed=11, kind="longrange")
    got = _host_generate(ld)
    assert got.dtype == ld.ld_data.dtype and np.array_equal(got, ld.ld_data)
    sk = syn.make_ld(sizes, low_memory=low_memory, ld_dtype=ld_dtype, seed=11, kind="longrange", data=False)
    assert sk.ld_data is None and sk.nnz == ld.nnz and sk.itemsize == ld.ld_data.dtype.itemsize
    assert np.array_equal(sk.ld_indptr, ld.ld_indptr) and np.array_equal(sk.ld_left_bound, ld.ld_left_bound)
    assert np.array_equal(_host_generate(sk), ld.ld_data)


def test_host_generator_cfg3_blocks_sample():
    """A few blocks of the genome-scale workload with the parameters bench.py draws for it (the largest one included)."""
    sizes_all = syn.block_sizes("cfg3")
    params_all = syn.longrange_params(sizes_all)
    rho_all = np.random.default_rng(syn.SEED + 1).uniform(0.3, 0.8, len(sizes_all))
    pick = [int(np.argmax(sizes_all)), 0, 7]
    for low_memory in (False, True):
        ld = syn.make_ld(sizes_all[pick], low_memory=low_memory, rho=rho_all[pick], kind="longrange",
                         params=[params_all[b] for b in pick])
        assert np.array_equal(_host_generate(ld), ld.ld_data)


def test_generator_rejects_bad_arguments():
    sizes = np.array([4, 0], dtype=np.int64)
    v = np.zeros(4, np.float32)
    out = np.zeros(64, np.float32)
    with pytest.raises(ValueError, match="block size"):
        L.check(L.lib.viprs_synthetic_ld_host(2, _ptr(sizes), _ptr(v), _ptr(v), _ptr(v), _ptr(v), _ptr(v), L.LD_F32, 0,
                                              _ptr(out), 64))
    sizes = np.array([4], dtype=np.int64)
    with pytest.raises(ValueError, match="too small"):
        L.check(L.lib.viprs_synthetic_ld_host(1, _ptr(sizes), _ptr(v), _ptr(v), _ptr(v), _ptr(v), _ptr(v), L.LD_F32, 0,
                                              _ptr(out), 15))
    with pytest.raises(ValueError, match="float32, int8 or int16"):
        L.check(L.lib.viprs_synthetic_ld_host(1, _ptr(sizes), _ptr(v), _ptr(v), _ptr(v), _ptr(v), _ptr(v), L.LD_F64, 0,
                                              _ptr(out), 64))


@pytest.mark.gpu
@pytest.mark.parametrize("low_memory", [False, True], ids=["sym", "upper"])
@pytest.mark.parametrize("ld_dtype", [np.float32, np.int8], ids=["f32", "i8"])
def test_generated_plan_equals_uploaded_plan(gpu, low_memory, ld_dtype):
    from viprs_amd.plan import DeviceState, LDPlan
    sizes = [1700, 64, 3, 700, 129]
    ld = syn.make_ld(sizes, low_memory=low_memory, ld_dtype=ld_dtype, seed=5, kind="longrange")
    ss = syn.make_sumstats(ld, seed=5)
    inp = syn.make_inputs(ss)
    sk = syn.make_ld(sizes, low_memory=low_memory, ld_dtype=ld_dtype, seed=5, kind="longrange", data=False)
    ref = H.run_oracle(ld, inp, inp.state_copy(), sweeps=2)
    outs = []
    for plan in (LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, low_memory), LDPlan.synthetic(sk)):
        assert plan.m == ld.m and plan.nnz == ld.nnz
        lb, ip = plan.windows()
        assert np.array_equal(lb, ld.ld_left_bound) and np.array_equal(ip, ld.ld_indptr)
        st = DeviceState(plan, "float32", "spike_slab", 1)
        for name in ("std_beta", "u_logs", "sqrt_half_var_tau", "mu_mult"):
            st.upload(name, getattr(inp, name))
        st.reset(inp.pi)
        for _ in range(2):
            st.e_step(ld.dq_scale, None, sync=True)
        outs.append({k: st.download(k) for k in H.STATE})
        st.close()
        plan.close()
    H.assert_state_equal(outs[0], ref)
    H.assert_state_equal(outs[1], ref)


@pytest.mark.gpu
def test_generated_cfg3_workload_equals_the_host_built_one(gpu):
    """bench.py's multi-GPU ranks (and its N = 1 secondaries) sweep LD that never existed on the host: at BASELINE configs[2]'s
    full size (1.1 M SNPs, 1 700 blocks, 3.8 GB) the generated plan and the plan uploaded from `synthetic.make_ld` leave the
    same state after a sweep, bit for bit -- and so do rank 0's blocks of an 8-way split."""
    import bench
    from viprs_amd.plan import DeviceState, LDPlan

    class A:
        math, ld_kind, host_ld = "exact", "longrange", False

    sizes = bench.config_sizes("cfg3", 7209)
    for mine in (None, bench.shard_blocks_lpt(sizes, 8)[0]):
        outs = []
        for data in (True, False):
            ld, ss, inp, m_all = bench.build_workload(A, sizes, mine, 7209, False, np.dtype("float32"), data=data)
            assert (ld.ld_data is None) == (not data) and m_all == int(sizes.sum())
            plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, False) if data else LDPlan.synthetic(ld)
            st = DeviceState(plan, "float32", "spike_slab", 1)
            for name in ("std_beta", "u_logs", "sqrt_half_var_tau", "mu_mult"):
                st.upload(name, getattr(inp, name))
            st.reset(inp.pi)
            st.e_step(ld.dq_scale, None, sync=True)
            outs.append({k: st.download(k) for k in H.STATE})
            st.close()
            plan.close()
            del ld
        H.assert_state_equal(outs[0], outs[1])
        assert np.count_nonzero(outs[0]["eta_diff"]) > 0.5 * outs[0]["eta_diff"].size
