"""GPU parity tests proper: the HIP E-step, called through the C ABI (ctypes shim with the
reference's `cpp_e_step` signature), against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

from tests import helpers as H
from viprs_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu


def _problem(sizes, low_memory, ld_dtype=np.float32, seed=11, kind="ar1"):
    return syn.make_problem(sizes=sizes, low_memory=low_memory, ld_dtype=ld_dtype, seed=seed, kind=kind)


@pytest.mark.parametrize("low_memory", [False, True])
@pytest.mark.parametrize("sizes", [[500], [37, 128, 300], [1, 2, 63, 64, 65, 129, 700], [1500, 90],
                                   [1700, 90], [2300, 1601, 1536, 1537, 40]])   # >= 1536: multi-CU teams
@pytest.mark.parametrize("kind", ["ar1", "longrange"])
def test_first_sweep_matches_oracle(gpu, sizes, low_memory, kind):
    ld, ss, inp = _problem(sizes, low_memory, kind=kind)
    st0 = inp.state_copy()
    ref = H.run_oracle(ld, inp, st0)
    got = H.run_hip(ld, inp, st0)
    H.assert_state_close(got, ref)
    H.assert_state_equal(got, ref)     # bit-for-bit in both LD forms


@pytest.mark.parametrize("low_memory", [False, True])
@pytest.mark.parametrize("kind", ["ar1", "longrange"])
def test_five_sweeps(gpu, low_memory, kind):
    ld, ss, inp = _problem([200, 333], low_memory, kind=kind)
    st0 = inp.state_copy()
    ref = H.run_oracle(ld, inp, st0, sweeps=5)
    got = H.run_hip(ld, inp, st0, sweeps=5)
    H.assert_state_close(got, ref)


@pytest.mark.parametrize("low_memory", [False, True])
@pytest.mark.parametrize("ld_dtype", [np.int8, np.int16])
@pytest.mark.parametrize("kind", ["ar1", "longrange"])
def test_quantised_ld(gpu, ld_dtype, low_memory, kind):
    ld, ss, inp = _problem([130, 257], low_memory, ld_dtype=ld_dtype, kind=kind)
    st0 = inp.state_copy()
    ref = H.run_oracle(ld, inp, st0, sweeps=2)
    got = H.run_hip(ld, inp, st0, sweeps=2)
    H.assert_state_close(got, ref)
    H.assert_state_equal(got, ref)
