"""Multi-GPU plumbing: one process per GPU, LD BLOCKS sharded over the ranks, and ONE small float64
collective per EM iteration for the M-step / ELBO sums.

The reference has no collectives at all (its only parallelism is OpenMP inside the kernel and joblib
over chromosomes, bin/viprs_fit:1080-1086); within one E-step call the hyper-parameters are fixed and
LD blocks share no ``q`` entries, so blocks are independent units and the data path needs no exchange.

* ``shard_blocks``     -- static, chain-aware longest-processing-time assignment of LD blocks to ranks;
* ``RcclComm``         -- the communicator of the GPU path: RCCL over xGMI through the C ABI
                          (``viprs_comm_*``, no PyTorch); attached to a ``DeviceState`` the device-resident
                          partial sums are all-gathered and reduced in rank order on the plan's stream;
* ``TorchDistComm``    -- ``torch.distributed`` process group; only the CPU test-suite uses it (``gloo``),
                          as the transport under the oracle-driven host logic;
* ``LocalComm``        -- single process.
"""
import ctypes
import os
import time

import numpy as np

# ---- block costs ------------------------------------------------------------------------------------
# A rank's sweep time is bounded by its HBM stream (sum of the LD bytes of its blocks), by the serial
# Gauss-Seidel chains of its blocks spread over the chain slots of the chip, and by the longest single chain.
HBM_STREAM_BYTES_PER_S = 5.5e12      # what a sweep streams at on one MI355X
CHAIN_STEP_S = 140e-9                # one serial SNP update of the panel kernels' chain wave, per-phase work included
CHAIN_SLOTS = 512                    # chains resident on one GPU (256 CUs x workgroups per CU)


def block_cost(size, elem_bytes=4, symmetric=True):
    """Additive cost (seconds) of one dense LD block of `size` SNPs: its share of the HBM stream or its
    share of the chip's chain slots, whichever is larger."""
    size = np.asarray(size, dtype=np.float64)
    stream = size * size * elem_bytes / HBM_STREAM_BYTES_PER_S
    chain = size * CHAIN_STEP_S / CHAIN_SLOTS
    return np.maximum(stream, chain)


def shard_blocks(sizes, world_size, elem_bytes=4):
    """Longest-processing-time assignment of LD blocks to ranks with the chain-aware cost.  Returns
    ``owner`` (one rank per block).  Deterministic and identical on every rank.  The longest chain of a
    rank (its largest block x CHAIN_STEP_S) is a floor no assignment can lower; LPT places the largest
    blocks first, one per rank, which is the best that can be done for it."""
    sizes = np.asarray(sizes, dtype=np.int64)
    cost = block_cost(sizes, elem_bytes)
    order = np.argsort(-cost, kind="stable")
    load = np.zeros(world_size)
    owner = np.zeros(len(sizes), dtype=np.int64)
    for i in order:
        r = int(np.argmin(load))
        owner[i] = r
        load[r] += cost[i]
    return owner


def rank_time_model(sizes, elem_bytes=4):
    """max(stream time, chain-slot time, longest chain) of one rank's blocks (seconds)."""
    sizes = np.asarray(sizes, dtype=np.float64)
    if sizes.size == 0:
        return 0.0
    return float(max((sizes * sizes).sum() * elem_bytes / HBM_STREAM_BYTES_PER_S,
                     sizes.sum() * CHAIN_STEP_S / CHAIN_SLOTS, sizes.max() * CHAIN_STEP_S))


class BlockShard:
    """The LD blocks of one chromosome that live on this rank: SNP index set + the LD rows re-indexed to it."""

    def __init__(self, block_start, mine):
        self.block_start = np.asarray(block_start, dtype=np.int64)
        self.blocks = np.asarray(sorted(int(b) for b in mine), dtype=np.int64)
        self.m_full = int(self.block_start[-1])
        if len(self.blocks):
            self.index = np.concatenate([np.arange(self.block_start[b], self.block_start[b + 1]) for b in self.blocks])
        else:
            self.index = np.zeros(0, dtype=np.int64)
        self.m = int(self.index.shape[0])

    def take(self, array):
        """Rows of a per-SNP array ((m_full,) or (m_full, k)) that belong to this rank."""
        return np.ascontiguousarray(np.asarray(array)[self.index])

    def scatter(self, local, full):
        full[self.index] = local
        return full

    def slice_ld(self, ld_left_bound, ld_indptr, ld_data):
        """(left_bound, indptr, data) of the local rows; windows shifted to the local SNP numbering.  Valid
        because a row's window never leaves its block (that is what makes it a block)."""
        lb_parts, len_parts, data_parts = [], [], []
        off = 0
        for b in self.blocks:
            s, e = int(self.block_start[b]), int(self.block_start[b + 1])
            lb_parts.append(ld_left_bound[s:e].astype(np.int64) - s + off)
            len_parts.append(np.diff(ld_indptr[s:e + 1]).astype(np.int64))
            data_parts.append(ld_data[int(ld_indptr[s]):int(ld_indptr[e])])
            off += e - s
        if not lb_parts:
            return (np.zeros(0, np.int32), np.zeros(1, ld_indptr.dtype), np.zeros(0, ld_data.dtype))
        lb = np.concatenate(lb_parts).astype(np.int32)
        ip = np.concatenate([[0], np.cumsum(np.concatenate(len_parts))]).astype(ld_indptr.dtype)
        return lb, ip, np.ascontiguousarray(np.concatenate(data_parts))

    def read_ld(self, ld_mat, dtype=None):
        """The same from a store that reads row ranges (`load_rows(start, stop, dtype)` ->
        (leftmost_idx, indptr re-based to 0, data)): only this rank's LD entries leave the disk."""
        lb_parts, len_parts, data_parts = [], [], []
        off = 0
        for b in self.blocks:
            s, e = int(self.block_start[b]), int(self.block_start[b + 1])
            lb, ip, data = ld_mat.load_rows(s, e, dtype)
            lb_parts.append(np.asarray(lb, dtype=np.int64) - s + off)
            len_parts.append(np.diff(ip).astype(np.int64))
            data_parts.append(data)
            off += e - s
        lb = np.concatenate(lb_parts).astype(np.int32)
        ip = np.concatenate([[0], np.cumsum(np.concatenate(len_parts))]).astype(np.int64)
        return lb, ip, np.ascontiguousarray(np.concatenate(data_parts))


# ---- communicators ------------------------------------------------------------------------------------
class LocalComm:
    """Single process: every reduction is the identity."""
    rank = 0
    world_size = 1
    device_side = False

    def allreduce_sum(self, vec):
        return np.asarray(vec, dtype=np.float64)

    def allreduce_max(self, vec):
        return np.asarray(vec, dtype=np.float64)

    def barrier(self):
        pass


_COMM_SEQ = 0


def _exchange_unique_id(rank, world_size, make_id, timeout_s=300.0):
    """Rank 0's RCCL unique id reaches the other ranks of the node through a file (atomic rename): no
    PyTorch, no extra port.  VIPRS_COMM_ID_FILE names it explicitly; by default it is keyed on the
    launcher (the parent process all ranks share under torch.distributed.run / a shell loop) and on
    MASTER_PORT."""
    global _COMM_SEQ
    path = os.environ.get("VIPRS_COMM_ID_FILE")
    if not path:
        tag = f"{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{os.getppid()}"
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"viprs_comm_{os.getuid()}_{tag}.id")
    # every communicator of a process gets its own file (all ranks create them in the same order): a rank can never
    # pick up the id of the previous communicator while rank 0 is still removing that file
    path = f"{path}.{_COMM_SEQ}"
    _COMM_SEQ += 1
    if rank == 0:
        uid = make_id()
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)
        return uid, path
    t0 = time.time()
    while True:
        try:
            with open(path, "rb") as f:
                uid = f.read()
            if len(uid) == 128:
                return uid, path
        except FileNotFoundError:
            pass
        if time.time() - t0 > timeout_s:
            raise TimeoutError(f"rank {rank}: no RCCL unique id at {path} after {timeout_s:.0f} s")
        time.sleep(0.01)


class RcclComm:
    """RCCL communicator over the C ABI (``viprs_comm_*``): one process per GPU of a node.

    ``device_side``: a ``DeviceState`` that this communicator is attached to (``DeviceState.set_comm``)
    returns ALL-RANK sums from ``sums_begin / sums_end`` -- the collective runs on the plan's stream."""
    device_side = True

    def __init__(self, rank=None, world_size=None, device=None):
        from . import _lib as L
        self._L = L
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world_size = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None else int(world_size)
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", str(self.rank))) % max(1, L.device_count())
        self.device = int(device)

        def make_id():
            buf = ctypes.create_string_buffer(L.COMM_ID_BYTES)
            L.check(L.lib.viprs_comm_unique_id(buf))
            return buf.raw

        if self.world_size > 1:
            uid, path = _exchange_unique_id(self.rank, self.world_size, make_id)
        else:
            uid, path = make_id(), None
        self._h = ctypes.c_void_p()
        L.check(L.lib.viprs_comm_create(ctypes.byref(self._h), uid, self.rank, self.world_size, self.device))
        if path is not None:
            self.barrier()                      # every rank has read the id: rank 0 may remove the file
            if self.rank == 0:
                try:
                    os.remove(path)
                except OSError:
                    pass

    @property
    def handle(self):
        if not self._h:
            raise ValueError("RcclComm is closed")
        return self._h

    def _reduce(self, vec, group):
        v = np.ascontiguousarray(vec, dtype=np.float64).copy()
        self._L.check(self._L.lib.viprs_comm_allreduce(self.handle, v.ctypes.data_as(ctypes.c_void_p),
                                                       int(v.size), int(group)))
        return v

    def allreduce_sum(self, vec):
        return self._reduce(vec, 0)

    def allreduce_max(self, vec):
        return self._reduce(vec, -1)

    def barrier(self):
        self._L.check(self._L.lib.viprs_comm_barrier(self.handle))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.lib.viprs_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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

    def barrier(self):
        self._dist.barrier()


class FileComm:
    """Host-vector collectives through files in a shared directory: a TEST transport that lets the multi-rank host
    logic (bench.py --gpus N, the models) run with several processes on a box whose GPUs RCCL cannot span (e.g. two
    ranks on one device).  Never the product path."""
    device_side = False

    def __init__(self, rank=None, world_size=None, root=None):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world_size = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None else int(world_size)
        tag = f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
        self.root = root or os.path.join(os.environ.get("TMPDIR", "/tmp"), f"viprs_filecomm_{os.getuid()}_{tag}")
        os.makedirs(self.root, exist_ok=True)
        self._n = 0

    def _exchange(self, vec):
        v = np.ascontiguousarray(vec, dtype=np.float64)
        self._n += 1
        mine = os.path.join(self.root, f"{self._n}_{self.rank}.npy")
        tmp = mine + ".tmp.npy"
        np.save(tmp, v)
        os.replace(tmp, mine)
        parts = []
        for r in range(self.world_size):
            f = os.path.join(self.root, f"{self._n}_{r}.npy")
            t0 = time.time()
            while not os.path.exists(f):
                if time.time() - t0 > 600:
                    raise TimeoutError(f"FileComm: rank {r} never wrote step {self._n}")
                time.sleep(0.002)
            parts.append(np.load(f))
        return np.stack(parts)

    def allreduce_sum(self, vec):
        return self._exchange(vec).sum(axis=0)

    def allreduce_max(self, vec):
        return self._exchange(vec).max(axis=0)

    def barrier(self):
        self._exchange(np.zeros(1))

    def close(self):
        pass


def broadcast_from_root(comm, values):
    """Rank 0's `values` (float64 vector) on every rank: an all-reduce of a vector that is zero elsewhere."""
    v = np.asarray(values, dtype=np.float64)
    if comm.world_size == 1:
        return v
    return comm.allreduce_sum(v if comm.rank == 0 else np.zeros_like(v))


def assign_chromosomes(costs, world_size):
    """Static longest-processing-time assignment of whole chromosomes (cost ~ LD entries) to ranks.
    Returns {chromosome: rank}; deterministic, identical on every rank.  (Kept for callers that want
    chromosome granularity; the models shard at LD-block granularity, see `shard_blocks`.)"""
    load = [0.0] * world_size
    owner = {}
    for c in sorted(costs, key=lambda k: (-costs[k], str(k))):
        r = min(range(world_size), key=lambda i: (load[i], i))
        owner[c] = r
        load[r] += float(costs[c])
    return owner
