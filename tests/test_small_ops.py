"""FFTShiftOperator, InversionEnabler and integer `uniform` draws against vectors generated from the real reference
(tests/golden/make_golden.py::small_ops_cases -> small_ops.npz), on the host and on the GPU (round 6)."""
import numpy as np
import pytest

import nifty_amd as ift
from tests import goldenlib as gl

SHIFT_CASES = {"a": (lambda: ift.RGSpace((8,)), None), "b": (lambda: ift.RGSpace((7, 6)), None),
               "c": (lambda: ift.DomainTuple.make((ift.RGSpace((4, 5)), ift.UnstructuredDomain(3), ift.RGSpace(6))), (0, -1)),
               "d": (lambda: ift.DomainTuple.make((ift.UnstructuredDomain(2), ift.RGSpace((5, 4, 3)))), 1)}


def _check_shift(tag, device_id):
    z = gl.load("small_ops")
    make, spaces = SHIFT_CASES[tag]
    dom = ift.DomainTuple.make(make())
    op = ift.FFTShiftOperator(dom, spaces)
    x = ift.makeField(dom, z[f"shift.{tag}.x"], device_id)
    xc = ift.makeField(dom, z[f"shift.{tag}.xc"], device_id)
    for got, key in ((op(x), "times"), (op.inverse(x), "inverse"), (op.adjoint(x), "adjoint"), (op(xc), "times_c")):
        assert got.device_id == device_id
        assert np.array_equal(got.asnumpy(), z[f"shift.{tag}.{key}"])  # a permutation: exact
    assert np.array_equal(op.adjoint.inverse(x).asnumpy(), z[f"shift.{tag}.times"])
    assert np.array_equal(op.inverse(op(x)).asnumpy(), z[f"shift.{tag}.x"])
    x32 = ift.makeField(dom, z[f"shift.{tag}.x"].astype(np.float32), device_id)
    assert np.array_equal(op(x32).asnumpy(), z[f"shift.{tag}.times"].astype(np.float32))


@pytest.mark.parametrize("tag", sorted(SHIFT_CASES))
def test_fftshift_host(tag):
    _check_shift(tag, -1)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(SHIFT_CASES))
def test_fftshift_device(tag):
    _check_shift(tag, 0)


def test_fftshift_argument_errors():
    sp = ift.RGSpace((4, 4))
    with pytest.raises(AssertionError):  # the reference asserts its arguments (harmonic_operators.py:408-415)
        ift.FFTShiftOperator(ift.UnstructuredDomain(5))
    with pytest.raises(AssertionError):
        ift.FFTShiftOperator(sp, spaces=3)
    op = ift.FFTShiftOperator(sp)
    with pytest.raises(ValueError):
        op(ift.full(ift.RGSpace((4, 5)), 1.0))
    assert op.capability == op.TIMES | op.ADJOINT_TIMES | op.INVERSE_TIMES | op.ADJOINT_INVERSE_TIMES


class _TimesOnly(ift.EndomorphicOperator):
    def __init__(self, op):
        self._op, self._domain, self._capability = op, op.domain, self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        return self._op.apply(x, mode)


def _check_inversion(device_id, tol):
    z = gl.load("small_ops")
    sp = ift.RGSpace((16, 12), (0.3, 0.2))
    HT = ift.HartleyOperator(sp)
    S = _TimesOnly(ift.SandwichOperator.make(HT, ift.DiagonalOperator(ift.makeField(HT.target, z["inv.diag"], device_id))))
    y = ift.makeField(sp, z["inv.y"], device_id)
    ic = lambda: ift.GradientNormController(tol_abs_gradnorm=1e-10, iteration_limit=40)  # noqa: E731
    plain = ift.InversionEnabler(S, ic())
    assert plain.capability == 15 and plain.domain is S.domain
    for mode in ("inverse_times", "adjoint_inverse_times"):
        got = getattr(plain, mode)(y)
        assert got.device_id == device_id
        assert gl.relerr(got.asnumpy(), z[f"inv.{mode}"]) < tol
    assert gl.relerr(plain(y).asnumpy(), z["inv.times"]) < 1e-13          # modes `op` has go straight through
    assert gl.relerr(S(plain.inverse_times(y)).asnumpy(), z["inv.y"]) < 1e-9
    approx = ift.ScalingOperator(sp, float(z["inv.approx_factor"]), np.float64)
    pre = ift.InversionEnabler(S, ift.GradientNormController(tol_abs_gradnorm=1e-10, iteration_limit=3), approximation=approx)
    assert gl.relerr(pre.inverse_times(y).asnumpy(), z["inv.preconditioned_3_steps"]) < tol  # 3 preconditioned CG steps
    with pytest.raises(NotImplementedError):
        plain.apply(y, 3)
    with pytest.raises(TypeError):
        ift.InversionEnabler(HT, ic())       # not endomorphic
    with pytest.raises(TypeError):
        ift.InversionEnabler(lambda x: x, ic())


def test_inversion_enabler_host():
    _check_inversion(-1, 1e-10)


@pytest.mark.gpu
def test_inversion_enabler_device():
    _check_inversion(0, 1e-10)
