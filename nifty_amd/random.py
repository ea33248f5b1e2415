"""Seed-sequence / generator STACK with the reference's discipline (nifty/cl/random.py:83-290).

Parity with the nifty.cl numpy path on identical seeds requires the very same numpy PCG64 streams,
spawned the same way, so the draws happen on the host with numpy and are uploaded to the device.
"""
import pickle

import numpy as np

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
