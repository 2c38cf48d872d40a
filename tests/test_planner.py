"""CPU: the host-side LD block planner (viprs_plan_blocks): bit-exact integer work."""
import numpy as np
import pytest

from viprs_amd import _lib as L
from viprs_amd.plan import plan_blocks
from viprs_amd.utils import synthetic as syn


def _components_numpy(lb, ip):
    """Independent model: merge the intervals ext(j) = [min(j, lb_j), max(j+1, lb_j+len_j))."""
    m = len(lb)
    lo = np.arange(m)
    hi = np.arange(m) + 1
    length = np.diff(ip)
    has = length > 0
    lo = np.where(has, np.minimum(lo, lb), lo)
    hi = np.where(has, np.maximum(hi, lb + length), hi)
    order = np.argsort(lo, kind="stable")
    starts = []
    cur_hi = -1
    for i in order:
        if lo[i] >= cur_hi:
            starts.append(int(lo[i]))
            cur_hi = int(hi[i])
        else:
            cur_hi = max(cur_hi, int(hi[i]))
    return np.array(starts + [m], dtype=np.int64)


@pytest.mark.parametrize("low_memory", [False, True])
@pytest.mark.parametrize("indptr_dtype", [np.int32, np.int64])
def test_block_ld_is_recovered_exactly(low_memory, indptr_dtype):
    sizes = [1, 2, 63, 64, 65, 500, 3, 129]
    ld = syn.make_ld(sizes, low_memory=low_memory, indptr_dtype=indptr_dtype)
    starts, kinds = plan_blocks(ld.ld_left_bound, ld.ld_indptr, low_memory)
    assert np.array_equal(starts, ld.block_start)
    want = L.BLOCK_DENSE_UPPER if low_memory else L.BLOCK_DENSE_SYM
    assert np.all(kinds == want)
    assert np.array_equal(starts, _components_numpy(ld.ld_left_bound.astype(np.int64), ld.ld_indptr.astype(np.int64)))


def test_banded_windows_form_one_ragged_component():
    m, w = 300, 20
    lb = np.maximum(np.arange(m) - w, 0).astype(np.int32)
    ub = np.minimum(np.arange(m) + w + 1, m)
    ip = np.concatenate([[0], np.cumsum(ub - lb)]).astype(np.int64)
    starts, kinds = plan_blocks(lb, ip, False)
    assert np.array_equal(starts, [0, m])
    assert kinds[0] == L.BLOCK_RAGGED


def test_random_windows_match_numpy_model():
    rng = np.random.default_rng(0)
    for _ in range(50):
        m = int(rng.integers(1, 200))
        lb = np.empty(m, dtype=np.int32)
        length = np.empty(m, dtype=np.int64)
        for j in range(m):
            lo = int(rng.integers(max(0, j - 6), j + 1))
            hi = int(rng.integers(j + 1, min(m, j + 7) + 1))
            if rng.random() < 0.1:
                lo, hi = j, j          # empty row
            lb[j], length[j] = lo, hi - lo
        ip = np.concatenate([[0], np.cumsum(length)]).astype(np.int64)
        starts, kinds = plan_blocks(lb, ip, False)
        assert np.array_equal(starts, _components_numpy(lb.astype(np.int64), ip))


def test_empty_and_validation_errors():
    starts, kinds = plan_blocks(np.zeros(0, np.int32), np.zeros(1, np.int64), False)
    assert np.array_equal(starts, [0]) and len(kinds) == 0
    lb = np.array([0, 0, 0], np.int32)
    with pytest.raises(L.ViprsLayoutError):                       # indptr[0] != 0
        plan_blocks(lb, np.array([1, 4, 7, 10], np.int64), False)
    with pytest.raises(L.ViprsLayoutError):                       # not monotone
        plan_blocks(lb, np.array([0, 3, 2, 5], np.int64), False)
    with pytest.raises(L.ViprsLayoutError):                       # window beyond m
        plan_blocks(lb, np.array([0, 3, 6, 10], np.int64), False)
    with pytest.raises(L.ViprsLayoutError):                       # negative left bound
        plan_blocks(np.array([-1, 0, 0], np.int32), np.array([0, 3, 6, 9], np.int64), False)
    with pytest.raises(ValueError):                               # the Cython boundary wants C int
        plan_blocks(lb.astype(np.int64), np.array([0, 3, 6, 9], np.int64), False)
    with pytest.raises(ValueError):
        plan_blocks(lb, np.array([0, 3, 6, 9], np.float64), False)


@pytest.mark.parametrize("low_memory", [False, True])
def test_merged_chromosomes_plan_is_the_union_of_the_per_chromosome_plans(low_memory):
    """VIPRS puts all local chromosomes into one device plan: the merged index arrays must describe
    exactly the per-chromosome LD blocks, shifted into place."""
    from viprs_amd.data import merge_ld_arrays
    from viprs_amd.plan import plan_blocks
    from viprs_amd.utils import synthetic as syn
    sizes = {20: [150, 90], 21: [1], 22: [64, 65, 333]}
    lds = {c: syn.make_ld(np.array(s), low_memory=low_memory, seed=40 + c) for c, s in sizes.items()}
    chroms = sorted(sizes)
    lb, ip, data, seg = merge_ld_arrays(chroms, {c: lds[c].m for c in chroms}, {c: lds[c].ld_left_bound for c in chroms},
                                        {c: lds[c].ld_indptr for c in chroms}, {c: lds[c].ld_data for c in chroms})
    assert lb.dtype == np.int32 and ip.dtype == np.int64 and ip[0] == 0 and ip[-1] == data.shape[0]
    starts, kinds = plan_blocks(lb, ip, low_memory)
    exp_starts, exp_kinds = [], []
    for c in chroms:
        st, kd = plan_blocks(lds[c].ld_left_bound, lds[c].ld_indptr, low_memory)
        exp_starts.append(st[:-1] + seg[c][0])
        exp_kinds.append(kd)
    assert np.array_equal(starts, np.concatenate(exp_starts + [[seg[chroms[-1]][1]]]))
    assert np.array_equal(kinds, np.concatenate(exp_kinds))
    for c in chroms:                                   # the data of every chromosome sits at its offset
        a, b = seg[c]
        assert np.array_equal(data[ip[a]:ip[b]], lds[c].ld_data)
