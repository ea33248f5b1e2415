"""Random numbers with the reference's seed discipline (nifty/cl/random.py:83-290): a STACK of numpy
SeedSequences, each with its own ``default_rng`` (PCG64) generator; draws always come from the top frame.

Parity with the nifty.cl numpy path on identical seeds requires the very same numpy streams, spawned the same way, so
the generators live on the host.  Large normal draws of device fields are computed on the GPU from the top generator's
state (``Random.normal_on_device`` -> nk_pcg64_normal): the values numpy would return -- bit-identical except in the
ziggurat tail |x| > 3.654 (2.7e-4 of the draws), where the device's log1p may differ from the host libm by <= 4 ulp
(tests/test_rng.py) -- and the host generator is advanced by exactly the raw draws consumed, so host and device draws
interleave like in the reference.
"""
import pickle
from collections import namedtuple

import numpy as np

# draws smaller than this stay on the host even for device fields (launch + sync cost more than numpy takes)
DEVICE_DRAW_MIN = 1 << 15

_Frame = namedtuple("_Frame", "sseq rng")


def _frame(sseq):
    return _Frame(sseq, np.random.default_rng(sseq))


_frames = [_frame(np.random.SeedSequence(42))]  # the reference starts every process on SeedSequence(42) too (:83-85)


class _StackView:
    """The reference keeps two parallel module lists `_sseq` / `_rng` (random.py:83-85); here they are views of one stack"""

    def __init__(self, field):
        self._field = field

    def __len__(self):
        return len(_frames)

    def __getitem__(self, i):
        return getattr(_frames[i], self._field)


_sseq, _rng = _StackView("sseq"), _StackView("rng")


# ---- the stack -----------------------------------------------------------------------------------------------------
def getState():
    """The whole stack as bytes, in the reference's own layout (random.py:88-96): the pickled pair
    (list of SeedSequences, list of Generators), bottom frame first -- so a `nifty_random_state` file written here is
    read by nifty.cl's `setState` and the other way round (resume file of optimize_kl, rank-synchronisation guard)."""
    return pickle.dumps(([f.sseq for f in _frames], [f.rng for f in _frames]))


def setState(state):
    """Restores what getState (ours or the reference's, random.py:98-110) returned.  Files of rounds 1-4 of this package
    held a list of (SeedSequence, Generator) frames instead; those are still read."""
    loaded = pickle.loads(state)
    if isinstance(loaded, tuple) and len(loaded) == 2 and all(isinstance(part, list) for part in loaded) \
            and len(loaded[0]) == len(loaded[1]) and all(isinstance(q, np.random.SeedSequence) for q in loaded[0]):
        loaded = list(zip(*loaded))
    elif not isinstance(loaded, list):
        raise TypeError("setState: not a random state of this module or of nifty.cl")
    frames = [_Frame(*f) for f in loaded]
    if not frames:
        raise ValueError("setState: empty generator stack")
    for f in frames:
        if not (isinstance(f.sseq, np.random.SeedSequence) and isinstance(f.rng, np.random.Generator)):
            raise TypeError("setState: not a random state of this module (expected (SeedSequence, Generator) frames)")
    _frames[:] = frames


def current_rng():
    return _frames[-1].rng


def spawn_sseq(n, parent=None):
    """`n` child SeedSequences of `parent` (default: the top of the stack); advances the parent's spawn counter."""
    return (_frames[-1].sseq if parent is None else parent).spawn(n)


def push_sseq(sseq):
    _frames.append(_frame(sseq))


def push_sseq_from_seed(seed):
    _frames.append(_frame(np.random.SeedSequence(seed)))


def pop_sseq():
    _frames.pop()


class Context:
    """``with Context(seed_or_sseq):`` -- draws inside the block come from a fresh generator on that seed; leaving the
    block restores the previous one and checks that the block left the stack balanced."""

    def __init__(self, inp):
        self._sseq = inp if isinstance(inp, np.random.SeedSequence) else np.random.SeedSequence(inp)
        self._height = None

    def __enter__(self):
        self._height = len(_frames)
        push_sseq(self._sseq)

    def __exit__(self, exc_type, exc_value, tb):
        pop_sseq()
        if len(_frames) != self._height:
            raise RuntimeError("inconsistent RNG usage detected")
        return exc_type is None


# ---- draws ---------------------------------------------------------------------------------------------------------
def _kind(dtype):
    dtype = np.dtype(dtype)
    for name, abstract in (("complex", np.complexfloating), ("float", np.floating), ("int", np.integer)):
        if np.issubdtype(dtype, abstract):
            return dtype, name
    return dtype, "other"


def _check_normal_args(kind, mean, std):
    if kind not in ("float", "complex"):
        raise TypeError("dtype must be float or complex")
    if not (np.isscalar(mean) and np.isscalar(std)):
        raise TypeError("mean and std must be scalars")
    if np.issubdtype(type(std), np.complexfloating):
        raise TypeError("std must not be complex")
    if kind == "float" and np.issubdtype(type(mean), np.complexfloating):
        raise TypeError("mean must not be complex for a real result field")
    if std < 0:
        raise ValueError("scale < 0")  # numpy's own message for rng.normal; the device path would accept it silently


def _two_parts(dtype, shape, draw):
    """complex result whose real and imaginary parts are drawn one after the other (real first: reference :226-230)."""
    out = np.empty(shape, dtype=dtype)
    out.real = draw(0)
    out.imag = draw(1)
    return out


class Random:
    """Static draw functions on the top generator (reference random.py:209-258)."""

    @staticmethod
    def normal(dtype, shape, mean=0.0, std=1.0):
        dtype, kind = _kind(dtype)
        _check_normal_args(kind, mean, std)
        rng = current_rng()
        if kind == "complex":
            centre = (np.real(mean), np.imag(mean))
            return _two_parts(dtype, shape, lambda part: rng.normal(centre[part], std, shape))
        return rng.normal(mean, std, shape).astype(dtype, copy=False)

    @staticmethod
    def normal_on_device(dtype, shape, mean, std, device):
        """`normal` for a field that lives on a GPU: the same numpy stream, drawn on the device (backend.pcg64_normal)
        unless the draw is small, of an exotic dtype, or config sampling_rng = "numpy_host"; a torch tensor on `device`."""
        import torch

        from . import backend, config
        from .field import _as_tensor

        dtype, kind = _kind(dtype)
        n = int(np.prod(shape, dtype=np.int64)) if len(tuple(shape)) else 1
        real_dt = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
                   np.dtype(np.complex64): torch.float32, np.dtype(np.complex128): torch.float64}.get(dtype)
        on_host = (config.get("sampling_rng") == "numpy_host" or n < DEVICE_DRAW_MIN or real_dt is None
                   or torch.device(device).type != "cuda")
        if on_host:
            return _as_tensor(Random.normal(dtype, shape, mean, std), device)
        _check_normal_args(kind, mean, std)
        rng = current_rng()
        if kind == "complex":
            re = backend.pcg64_normal(rng, np.real(mean), std, shape, real_dt, device)
            return torch.complex(re, backend.pcg64_normal(rng, np.imag(mean), std, shape, real_dt, device))
        return backend.pcg64_normal(rng, mean, std, shape, real_dt, device)

    @staticmethod
    def on_device(random_type, dtype, shape, device, **kwargs):
        """A draw of `random_type` for a field that lives on a GPU: the same numpy stream computed there where a device
        kernel exists (normal; uniform and pm1 of float / complex fields), else drawn on the host and uploaded.  Small draws
        stay on the host (a launch and its bookkeeping cost more than numpy takes)."""
        import torch

        from . import backend, config
        from .field import _as_tensor

        dt, kind = _kind(dtype)
        n = int(np.prod(shape, dtype=np.int64)) if len(tuple(shape)) else 1
        real_dt = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
                   np.dtype(np.complex64): torch.float32, np.dtype(np.complex128): torch.float64}.get(dt)
        on_gpu = config.get("sampling_rng") != "numpy_host" and n >= DEVICE_DRAW_MIN and torch.device(device).type == "cuda"
        host = not on_gpu or real_dt is None
        if random_type == "uniform" and kind == "int" and on_gpu and set(kwargs) <= {"low", "high"}:
            # integer fields: numpy's bounded-integer stream (Lemire's method, rejected words skipped) on the device
            low, high = kwargs.get("low", 0), kwargs.get("high", 1)
            if not (np.issubdtype(type(low), np.integer) and np.issubdtype(type(high), np.integer)):
                raise TypeError("low and high must be integer")
            from .field import _NP2T
            return backend.pcg64_integers(current_rng(), int(low), int(high), shape, device).to(_NP2T.get(dt, torch.int64))
        if random_type == "normal":
            return Random.normal_on_device(dtype, shape, kwargs.pop("mean", 0.0), kwargs.pop("std", 1.0), device, **kwargs)
        rng = current_rng()
        if random_type == "pm1" and not host and not kwargs:
            return backend.pcg64_pm1(rng, shape, real_dt, device, complex_units=kind == "complex")
        if random_type == "uniform" and not host and set(kwargs) <= {"low", "high"}:
            low, high = kwargs.get("low", 0.0), kwargs.get("high", 1.0)
            if not (np.isscalar(low) and np.isscalar(high)):
                raise TypeError("low and high must be scalars")
            if kind == "complex":  # real part first, then the imaginary part, like the host draw
                re = backend.pcg64_uniform(rng, low, high, shape, real_dt, device)
                return torch.complex(re, backend.pcg64_uniform(rng, low, high, shape, real_dt, device))
            return backend.pcg64_uniform(rng, low, high, shape, real_dt, device)
        return _as_tensor(getattr(Random, random_type)(dtype=dtype, shape=shape, **kwargs), device)

    @staticmethod
    def pm1(dtype, shape):
        """+-1 (real) or one of 1, i, -1, -i (complex) with equal probability."""
        dtype, kind = _kind(dtype)
        # the n-th roots of unity for n = 4 (complex) or 2 (real), picked by ONE bounded integer draw per element -- the
        # reference's `integers(0, n)` call, so the generator advances identically
        roots = np.array([1, 1j, -1, -1j] if kind == "complex" else [-1, 1], dtype=dtype)
        return roots[current_rng().integers(0, len(roots), size=shape)]

    @staticmethod
    def uniform(dtype, shape, low=0.0, high=1.0):
        """Uniform on [low, high): floats; integers: low..high inclusive; complex: both parts."""
        dtype, kind = _kind(dtype)
        if not (np.isscalar(low) and np.isscalar(high)):
            raise TypeError("low and high must be scalars")
        rng = current_rng()
        if kind == "complex":
            return _two_parts(dtype, shape, lambda part: rng.uniform(low, high, shape))
        if kind == "int":
            return rng.integers(low, high + 1, shape).astype(dtype, copy=False)
        return rng.uniform(low, high, shape).astype(dtype, copy=False)
