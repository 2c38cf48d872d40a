"""RCCL communicator smoke test under a launcher (RANK / LOCAL_RANK / WORLD_SIZE in the environment):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29511 tools/comm_smoke.py
(on a 1-GPU box every rank lands on device 0; RCCL may refuse that -- NCCL_DEBUG=WARN shows why)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from viprs_amd.parallel import RcclComm
c = RcclComm()
v = c.allreduce_sum(np.array([1.0, float(c.rank)]))
m = c.allreduce_max(np.array([float(c.rank)]))
c.barrier()
print(f"rank {c.rank}/{c.world_size} device {c.device}: sum {v} max {m}", flush=True)
assert v[0] == c.world_size and m[0] == c.world_size - 1
c.close()
