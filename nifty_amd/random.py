"""Seed-sequence / generator STACK with the reference's discipline (nifty/cl/random.py:83-290).

Parity with the nifty.cl numpy path on identical seeds requires the very same numpy PCG64 streams,
spawned the same way: the generators live on the host (numpy); large normal draws of device fields are computed on the
GPU from the generator's state (`Random.normal_on_device` -> nk_pcg64_normal, the same numbers draw for draw).
"""
import pickle

import numpy as np

# draws smaller than this stay on the host even for device fields (launch + sync cost more than numpy takes)
DEVICE_DRAW_MIN = 1 << 15

_sseq = [np.random.SeedSequence(42)]
_rng = [np.random.default_rng(_sseq[-1])]


def getState():
    return pickle.dumps((_sseq, _rng))


def setState(state):
    global _sseq, _rng
    _sseq, _rng = pickle.loads(state)


def spawn_sseq(n, parent=None):
    if parent is None:
        parent = _sseq[-1]
    return parent.spawn(n)


def current_rng():
    return _rng[-1]


def push_sseq(sseq):
    _sseq.append(sseq)
    _rng.append(np.random.default_rng(_sseq[-1]))


def push_sseq_from_seed(seed):
    push_sseq(np.random.SeedSequence(seed))


def pop_sseq():
    _sseq.pop()
    _rng.pop()


class Random:
    @staticmethod
    def normal(dtype, shape, mean=0.0, std=1.0):
        dtype = np.dtype(dtype)
        if not (np.issubdtype(dtype, np.floating) or np.issubdtype(dtype, np.complexfloating)):
            raise TypeError("dtype must be float or complex")
        if not np.isscalar(mean) or not np.isscalar(std):
            raise TypeError("mean and std must be scalars")
        if np.issubdtype(type(std), np.complexfloating):
            raise TypeError("std must not be complex")
        if np.issubdtype(dtype, np.complexfloating):
            x = np.empty(shape, dtype=dtype)
            x.real = _rng[-1].normal(np.real(mean), std, shape)
            x.imag = _rng[-1].normal(np.imag(mean), std, shape)
            return x
        if np.issubdtype(type(mean), np.complexfloating):
            raise TypeError("mean must not be complex for a real result field")
        return _rng[-1].normal(mean, std, shape).astype(dtype, copy=False)

    @staticmethod
    def normal_on_device(dtype, shape, mean, std, device):
        """`normal` for a field that lives on a GPU: the same numpy stream, drawn on the device (backend.pcg64_normal)
        unless the draw is small or config sampling_rng = "numpy_host"; returns a torch tensor on `device`."""
        import torch

        from . import backend, config
        from .field import _as_tensor

        dtype = np.dtype(dtype)
        n = int(np.prod(shape, dtype=np.int64)) if len(tuple(shape)) else 1
        if config.get("sampling_rng") == "numpy_host" or n < DEVICE_DRAW_MIN or torch.device(device).type != "cuda":
            return _as_tensor(Random.normal(dtype, shape, mean, std), device)
        if not (np.issubdtype(dtype, np.floating) or np.issubdtype(dtype, np.complexfloating)):
            raise TypeError("dtype must be float or complex")
        if not np.isscalar(mean) or not np.isscalar(std):
            raise TypeError("mean and std must be scalars")
        if np.issubdtype(type(std), np.complexfloating):
            raise TypeError("std must not be complex")
        if np.issubdtype(dtype, np.complexfloating):
            rdt = torch.float32 if dtype == np.complex64 else torch.float64
            re = backend.pcg64_normal(_rng[-1], np.real(mean), std, shape, rdt, device)
            im = backend.pcg64_normal(_rng[-1], np.imag(mean), std, shape, rdt, device)
            return torch.complex(re, im)
        if np.issubdtype(type(mean), np.complexfloating):
            raise TypeError("mean must not be complex for a real result field")
        tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}.get(dtype)
        if tdt is None:  # float16 and friends: host path
            return _as_tensor(Random.normal(dtype, shape, mean, std), device)
        return backend.pcg64_normal(_rng[-1], mean, std, shape, tdt, device)

    @staticmethod
    def pm1(dtype, shape):
        dtype = np.dtype(dtype)
        if np.issubdtype(dtype, np.complexfloating):
            x = np.array([1 + 0j, 0 + 1j, -1 + 0j, 0 - 1j], dtype=dtype)
            x = x[_rng[-1].integers(0, 4, size=shape)]
        else:
            x = 2 * _rng[-1].integers(0, 2, size=shape) - 1
        return x.astype(dtype, copy=False)

    @staticmethod
    def uniform(dtype, shape, low=0.0, high=1.0):
        dtype = np.dtype(dtype)
        if not np.isscalar(low) or not np.isscalar(high):
            raise TypeError("low and high must be scalars")
        if np.issubdtype(dtype, np.complexfloating):
            x = np.empty(shape, dtype=dtype)
            x.real = _rng[-1].uniform(low, high, shape)
            x.imag = _rng[-1].uniform(low, high, shape)
        elif np.issubdtype(dtype, np.integer):
            x = _rng[-1].integers(low, high + 1, shape)
        else:
            x = _rng[-1].uniform(low, high, shape)
        return x.astype(dtype, copy=False)


class Context:
    """``with Context(seed_or_sseq):`` pushes a fresh generator and pops it on exit."""

    def __init__(self, inp):
        if not isinstance(inp, np.random.SeedSequence):
            inp = np.random.SeedSequence(inp)
        self._sseq = inp

    def __enter__(self):
        self._depth = len(_sseq)
        push_sseq(self._sseq)

    def __exit__(self, exc_type, exc_value, tb):
        pop_sseq()
        if self._depth != len(_sseq):
            raise RuntimeError("inconsistent RNG usage detected")
        return exc_type is None
