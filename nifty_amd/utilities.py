"""The helpers user scripts and the reference's own tests reach through ``ift.utilities`` (reference nifty/cl/utilities.py), under
the reference's names -- most of them live elsewhere in this package (parallel.py, engine.py) and are gathered here.  The
"MPI" of this package is torch.distributed: one process per GPU, RCCL between them."""
import numpy as np
import torch

from . import parallel
from .parallel import (check_MPI_equality, check_MPI_synced_random_state, ensure_all_tasks_succeed,  # noqa: F401
                       get_MPI_params_from_comm, shareRange)


def get_MPI_params():
    """(comm, ntask, rank, master) of the running script (utilities.py:317-346): the world communicator when
    torch.distributed is initialised with more than one rank, else (None, 1, 0, True)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        comm = parallel.Comm()
        return comm, comm.size, comm.rank, comm.rank == 0
    return None, 1, 0, True


def allreduce_sum(obj, comm):
    """Sum of the tasks' lists of terms in the order of the reference's pairwise tree over the GLOBAL index, so that the result
    does not depend on the number of tasks (utilities.py:349-414).  Without a communicator: the tree over the local list."""
    terms = list(obj)
    if comm is None or comm.size == 1:
        if not terms:
            raise RuntimeError("empty operand list")
        return parallel.tree_fold(terms)
    import types

    from .kl import SampleListBase  # the distributed tree over Fields / MultiFields / floats lives with the sample lists

    # a rank without terms receives the total on the host
    holder = types.SimpleNamespace(_comm=comm, _device_id=lambda: getattr(terms[0], "device_id", -1) if terms else -1)
    return SampleListBase._sum_over_ranks(holder, terms)


def lognormal_moments(mean, sigma, N=0):
    """(logmean, logsigma) of the normal distribution whose exponential has the given mean and standard deviation
    (utilities.py:500-513); scalars or arrays of length N."""
    def shaped(x):
        x = np.asarray(x, dtype=float)
        if x.shape in ((), (1,)):
            return np.full(N, x) if N != 0 else x.reshape(())
        if x.shape != (N,):
            raise TypeError("x and N are incompatible")
        return x

    mean, sigma = shaped(mean), shaped(sigma)
    for name, val in (("mean", mean), ("sig", sigma)):
        if not np.all(val > 0):
            raise ValueError(f"{name} must be greater 0; got {val!r}")
    logsigma = np.sqrt(np.log1p((sigma / mean) ** 2))
    return np.log(mean) - 0.5 * logsigma ** 2, logsigma


def myassert(val):
    """An assert that survives ``python -O`` (utilities.py:516-520)."""
    if not val:
        raise AssertionError


def check_object_identity(obj0, obj1):
    if obj0 is not obj1:
        raise ValueError(f"Mismatch:\n{obj0}\n{obj1}")


def _one_answer(fn, values):
    answers = {fn(v) for v in values}
    if len(answers) != 1:
        raise RuntimeError("Value is not unique", sorted(answers))
    return answers.pop()


def iscomplextype(dtype):
    """True for complex dtypes; a dict of dtypes must agree (utilities.py:250-253)."""
    if isinstance(dtype, dict):
        return _one_answer(iscomplextype, dtype.values())
    return bool(np.issubdtype(dtype, np.complexfloating))


def issingleprec(dtype):
    if isinstance(dtype, dict):
        return _one_answer(issingleprec, dtype.values())
    return np.dtype(dtype).type in (np.float32, np.complex64)


def my_sum(iterable):
    total = None
    for term in iterable:
        total = term if total is None else total + term
    return total


def my_product(iterable):
    total = None
    for term in iterable:
        total = term if total is None else total * term
    return total


def indent(inp):
    return "\n".join(("  " + line).rstrip() for line in inp.splitlines())


def device_available():
    """A GPU this package can compute on (the reference asks cupy; here: PyTorch-ROCm + the built libniftyk)."""
    return bool(torch.cuda.is_available())


def assert_device_available():
    if not device_available():
        raise RuntimeError("no GPU device available")
