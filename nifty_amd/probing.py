"""Sample statistics and the sampled diagonal approximation of a metric (reference nifty/cl/probing.py:24-75, 142-152):
the `napprox` preconditioner of the MGVI sampling solves and of NewtonCG."""
import numpy as np

from .field import Field, MultiField


class StatCalculator:
    """Running mean and unbiased variance of the values added so far (Welford update; constant memory).  Works on
    anything with scalar multiplication and element-wise + - *: Fields, MultiFields, floats."""

    def __init__(self):
        self._count = 0

    def add(self, value):
        """Welford update: mean and the sum of squared deviations M2 advance together, in one pass over `value`."""
        first = self._count == 0
        self._count += 1
        if first:
            self._mean, self._m2 = value * 1.0, value * 0.0
        else:
            before = value - self._mean                      # deviation from the old mean ...
            self._mean = self._mean + before * (1.0 / self._count)
            self._m2 = self._m2 + before * (value - self._mean)  # ... times the deviation from the new one

    @property
    def mean(self):
        if self._count == 0:
            raise RuntimeError("no samples yet")
        return self._mean * 1.0

    @property
    def var(self):
        if self._count < 2:
            raise RuntimeError("need at least two samples")
        return self._m2 * (1.0 / (self._count - 1))


def approximation2endo(op, nsamples, device_id=-1):
    """Diagonal approximation of a metric: the variance of `nsamples` draws from it, zeros replaced by ones
    (probing.py:142-152).  Returned as a (Multi)Field on op.domain, on the device the draws were made on."""
    sc = StatCalculator()
    for _ in range(nsamples):
        sc.add(op.draw_sample(device_id=device_id))
    approx = sc.var

    def fix(f):
        zero = f.asnumpy() == 0  # host mask (the reference edits a host copy as well)
        if not zero.any():
            return f
        arr = f.asnumpy().copy()
        arr[zero] = 1
        return Field(f.domain, arr).at(f.device_id)

    if isinstance(approx, MultiField):
        return MultiField.from_dict({k: fix(approx[k]) for k in approx.keys()}, approx.domain)
    return fix(approx)
