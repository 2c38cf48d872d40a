"""Multi-GPU plumbing: one process per GPU, LD blocks (whole chromosomes) sharded over the ranks,
and ONE small float64 all-reduce per EM iteration for the M-step / ELBO sums.

The reference has no collectives at all (its only parallelism is OpenMP inside the kernel and
joblib over chromosomes, bin/viprs_fit:1080-1086); within one E-step call the hyper-parameters
are fixed, so blocks are independent and the data path needs no exchange.  ``torch.distributed``
is used purely as the transport: backend ``nccl`` is RCCL over xGMI on ROCm, ``gloo`` serves the
CPU tests.
"""
import numpy as np


class LocalComm:
    """Single process: the all-reduce is the identity."""
    rank = 0
    world_size = 1

    def allreduce_sum(self, vec):
        return np.asarray(vec, dtype=np.float64)

    def allreduce_max(self, vec):
        return np.asarray(vec, dtype=np.float64)

    def barrier(self):
        pass


class TorchDistComm:
    """``torch.distributed`` process group (already initialised by the launcher/torchrun)."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._torch, self._dist = torch, dist
        self.rank = dist.get_rank()
        self.world_size = dist.get_world_size()
        if device is None:
            device = "cuda" if dist.get_backend() == "nccl" else "cpu"
        self.device = device

    def _reduce(self, vec, op):
        t = self._torch.as_tensor(np.asarray(vec, dtype=np.float64)).to(self.device)
        self._dist.all_reduce(t, op=op)
        return t.cpu().numpy()

    def allreduce_sum(self, vec):
        return self._reduce(vec, self._dist.ReduceOp.SUM)

    def allreduce_max(self, vec):
        return self._reduce(vec, self._dist.ReduceOp.MAX)

    def barrier(self):
        self._dist.barrier()


def assign_chromosomes(costs, world_size):
    """Static longest-processing-time assignment of chromosomes (cost ~ LD entries) to ranks.
    Returns {chromosome: rank}; deterministic, identical on every rank."""
    load = [0.0] * world_size
    owner = {}
    for c in sorted(costs, key=lambda k: (-costs[k], str(k))):
        r = min(range(world_size), key=lambda i: (load[i], i))
        owner[c] = r
        load[r] += float(costs[c])
    return owner
