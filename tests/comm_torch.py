"""``TorchDistComm`` -- a ``viprs_amd.parallel`` communicator over a ``torch.distributed`` process group.  TEST transport
only (``gloo`` on CPU, world_size >= 2): it drives the multi-rank host logic of the models with the oracle's kernels.  The
product's multi-GPU path is ``viprs_amd.parallel.RcclComm`` (RCCL through the C ABI); the package itself never imports
torch."""
import numpy as np


class TorchDistComm:
    """``torch.distributed`` process group (already initialised by the launcher).  CPU-test transport
    (``gloo``) of the oracle-driven host logic; the GPU path uses ``RcclComm``."""
    device_side = False

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
        t = self._torch.as_tensor(np.array(vec, dtype=np.float64)).to(self.device)
        self._dist.all_reduce(t, op=op)
        return t.cpu().numpy()

    def allreduce_sum(self, vec):
        return self._reduce(vec, self._dist.ReduceOp.SUM)

    def allreduce_max(self, vec):
        return self._reduce(vec, self._dist.ReduceOp.MAX)

    def allgather(self, vec):
        t = self._torch.as_tensor(np.array(vec, dtype=np.float64)).to(self.device)
        parts = [self._torch.empty_like(t) for _ in range(self.world_size)]
        self._dist.all_gather(parts, t)
        return np.stack([p.cpu().numpy() for p in parts])

    def barrier(self):
        self._dist.barrier()
