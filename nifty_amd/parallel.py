"""Sample-parallel data parallelism: one process per GPU, torch.distributed over RCCL/xGMI.

Counterpart of the reference's only parallel strategy: samples are partitioned by
``shareRange`` (nifty/cl/utilities.py:282-306) and per-sample contributions are summed over ranks by
``allreduce_sum`` (utilities.py:349-414, called from sample_list.py:237,265).  Here the sum is ONE
in-place all-reduce per evaluation over the device-resident latent vector (backend "nccl" = RCCL on
ROCm; "gloo" for the CPU tests).  Unlike the reference's pairwise host tree the RCCL reduction order
depends on the rank count, i.e. results agree between rank counts to rounding (not bitwise).
"""
import os

import torch
import torch.distributed as dist


def shareRange(nwork, nshares, myshare):
    """Contiguous, as-even-as-possible block of work items for ``myshare`` (utilities.py:282-306)."""
    nbase, additional = divmod(int(nwork), int(nshares))
    lo = myshare * nbase + min(myshare, additional)
    return lo, lo + nbase + int(myshare < additional)


class Comm:
    """Minimal communicator facade (rank, size, in-place sum, broadcast, barrier)."""

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised; call parallel.init() first")
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.backend = str(dist.get_backend(group))
        self.backend_is_nccl = "nccl" in self.backend

    def allreduce_sum_(self, tensors):
        """In-place sum over ranks of every tensor in the list (one collective per tensor)."""
        for t in tensors:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return tensors

    def bcast_(self, tensor, root=0):
        dist.broadcast(tensor, src=root, group=self.group)
        return tensor

    def barrier(self):
        dist.barrier(group=self.group)

    def max_float(self, value, device):
        t = torch.tensor([float(value)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    @property
    def is_master(self):
        return self.rank == 0


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK/WORLD_SIZE/MASTER_*).

    Returns (comm or None, local_rank).  With WORLD_SIZE unset or 1 no process group is created.
    """
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world <= 1:
        return None, local_rank
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if not dist.is_initialized():
        dist.init_process_group(backend=backend)
    return Comm(), local_rank


def get_MPI_params_from_comm(comm):
    """(ntask, rank, master) like the reference's helper of the same name (utilities.py:309-316)."""
    if comm is None:
        return 1, 0, True
    return comm.size, comm.rank, comm.rank == 0
