"""GPU: the RCCL communicator of the C ABI (viprs_comm_*) -- world size 1 on the GPU box (8-GPU runs are the
driver's), so the collective path itself (dlopen of librccl, ncclCommInitRank, the all-gather + rank-ordered
reduction on the plan's stream, the empty-plan participation) is exercised end to end."""
import numpy as np
import pytest

from tests import helpers as H
from viprs_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def comm(gpu):
    from viprs_amd.parallel import RcclComm
    c = RcclComm(rank=0, world_size=1, device=0)
    yield c
    c.close()


def test_host_vector_collectives(comm):
    assert comm.size() == 1                      # what RCCL itself reports (ncclCommCount), bench.py's `rccl_ranks`
    v = np.array([1.5, -2.0, 3.25, 7.0])
    np.testing.assert_array_equal(comm.allreduce_sum(v), v)
    np.testing.assert_array_equal(comm.allreduce_max(v), v)
    comm.barrier()
    big = np.arange(100_000, dtype=np.float64)
    np.testing.assert_array_equal(comm.allreduce_sum(big), big)
    # bulk exchange (the posterior of every rank's SNPs at the end of a fit): (world, n) rows in rank order
    g = comm.allgather(big)
    assert g.shape == (1, big.size)
    np.testing.assert_array_equal(g[0], big)


def test_device_sums_through_the_communicator_equal_local_sums(comm):
    from viprs_amd.plan import DeviceState, LDPlan
    ld, ss, inp = syn.make_problem(sizes=[300, 1400, 77], low_memory=False, seed=4)
    plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, False)
    st = DeviceState(plan)
    st.upload("std_beta", inp.std_beta)
    st.set_n_per_snp(ss.n_per_snp)
    st.reset(inp.pi)
    st.prep(float(np.log(inp.pi) - np.log(1 - inp.pi)), float(np.log(inp.tau_beta)), inp.sigma_epsilon, inp.tau_beta, 1.0)
    st.e_step(1.0)
    local = st.sums(1.0)
    st.set_comm(comm)
    st.sums_begin(1.0)
    both = st.sums_end()
    np.testing.assert_array_equal(both, local)                   # one rank: the ordered reduction is the identity
    st.set_comm(None)
    st.close()
    plan.close()
    # a rank whose plan is empty still takes part in the collective and contributes zeros
    z = lambda dt=np.float32: np.zeros(0, dt)
    empty = LDPlan(z(np.int32), np.zeros(1, np.int64), z(), False)
    es = DeviceState(empty)
    es.set_comm(comm)
    es.sums_begin(1.0)
    np.testing.assert_array_equal(es.sums_end(), np.zeros(11))
    # ... also through the blocking entry points (they used to return before the collective, which
    # leaves the ranks that do hold blocks waiting in the all-gather)
    np.testing.assert_array_equal(es.sums(1.0), np.zeros(11))
    es.close()
    eg = DeviceState(empty, "float32", "grid", 3)
    eg.set_comm(comm)
    np.testing.assert_array_equal(eg.sums_column(1, 1.0), np.zeros(11))
    eg.close()
    empty.close()


def test_fit_with_rccl_comm_reproduces_the_single_process_fit(comm):
    """VIPRS(comm=RcclComm) -- device-resident iteration, merged plan, sums reduced by the communicator on the
    plan's stream -- against the plain single-process fit (same trajectory, bit for bit at world size 1)."""
    from viprs_amd.data import ArrayDataLoader
    from viprs_amd.model import VIPRS
    gdl = ArrayDataLoader.synthetic({1: [200, 90, 310], 2: [150, 260]}, seed=21)
    theta = {"pi": 0.02, "sigma_epsilon": 0.85}
    a = VIPRS(gdl, low_memory=True).fit(max_iter=15, theta_0=dict(theta))
    b = VIPRS(gdl, low_memory=True, comm=comm).fit(max_iter=15, theta_0=dict(theta))
    np.testing.assert_array_equal(a.history["ELBO"], b.history["ELBO"])
    for c in a.chromosomes:
        np.testing.assert_array_equal(a.pip[c], b.pip[c])


@pytest.mark.parametrize("family", ["spike_slab", "mixture"])
def test_per_chromosome_batch_with_rccl_comm(comm, family):
    """The lock-step batch of per-chromosome models with the groups' sums reduced by the communicator on the plan's stream (one
    all-gather for all groups: `viprs_state_sums_groups_*` / `viprs_state_sums_mixture_groups_*` with `viprs_state_set_comm`)
    against the batch without a communicator: the same trajectories, bit for bit at world size 1."""
    from viprs_amd.data import ArrayDataLoader
    from viprs_amd.model import VIPRSMixPerChromosome, VIPRSPerChromosome
    gdl = ArrayDataLoader.synthetic({1: [200, 90, 310], 2: [150, 260], 3: [330]}, seed=23)
    if family == "mixture":
        cls, kw, theta = VIPRSMixPerChromosome, {"K": 3}, {"pis": np.array([0.01, 0.005, 0.002]), "sigma_epsilon": 0.85}
    else:
        cls, kw, theta = VIPRSPerChromosome, {}, {"pi": 0.02, "sigma_epsilon": 0.85}
    a = cls(gdl, low_memory=True, **kw).fit(max_iter=15, theta_0=dict(theta))
    b = cls(gdl, low_memory=True, comm=comm, **kw).fit(max_iter=15, theta_0=dict(theta))
    assert b._device_reduce
    for c in a.groups:
        np.testing.assert_array_equal(a.history[c]["ELBO"], b.history[c]["ELBO"])
        np.testing.assert_array_equal(a.pip[c], b.pip[c])
        assert a.optim_results[c].nit == b.optim_results[c].nit
