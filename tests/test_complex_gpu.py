"""Complex fields on the device (VERDICT r3 "missing" item 3): element-wise products / quotients, DiagonalOperator with a
complex diagonal in all four modes (reference diagonal_operator.py:194-214), complex pointwise functions and Gaussian
energies of complex residuals (energy_operators.py:517-595) -- device results against the host path of the same classes."""
import numpy as np
import pytest
import torch

import nifty_amd as ift

pytestmark = pytest.mark.gpu


def _cfield(dom, rng, dtype=np.complex128):
    return ift.makeField(dom, (rng.normal(size=dom.shape) + 1j * rng.normal(size=dom.shape)).astype(dtype))


def _close(dev, host, tol):
    a, b = dev.asnumpy(), host.asnumpy()
    assert a.dtype == b.dtype and np.max(np.abs(a - b)) <= tol * np.max(np.abs(b))


@pytest.mark.parametrize("dtype,tol", [(np.complex128, 1e-14), (np.complex64, 2e-6)])
def test_complex_field_arithmetic_on_the_device(dtype, tol):
    rng = np.random.default_rng(1)
    dom = ift.DomainTuple.make(ift.RGSpace((6, 10)))
    a, b = _cfield(dom, rng, dtype), _cfield(dom, rng, dtype)
    r = ift.makeField(dom, rng.normal(size=dom.shape).astype(np.float64 if dtype == np.complex128 else np.float32))
    ad, bd, rd = a.at(0), b.at(0), r.at(0)
    for dev, host in ((ad * bd, a * b), (ad / bd, a / b), (ad + bd, a + b), (ad - bd, a - b), (ad * rd, a * r), (rd * ad, r * a),
                      (ad / rd, a / r), (ad * (2.0 - 0.5j), a * (2.0 - 0.5j)), (ad * 3.0, a * 3.0), (ad + (1.0 + 2.0j), a + (1.0 + 2.0j)),
                      (ad.conjugate(), a.conjugate()), (ad.ptw("exp"), a.ptw("exp")), (ad.ptw("reciprocal"), a.ptw("reciprocal")),
                      (ad.ptw("sqrt"), a.ptw("sqrt")), (ad.ptw("log"), a.ptw("log"))):
        assert dev.device_id == 0
        _close(dev, host, tol)
    # modulus: a real field
    m = abs(ad)
    assert not m.val.is_complex()
    assert np.max(np.abs(m.asnumpy() - np.abs(a.asnumpy()))) <= tol * np.max(np.abs(a.asnumpy()))
    # complex scalar product conj(a).b
    got, ref = ad.s_vdot(bd), a.s_vdot(b)
    assert abs(got - ref) <= 10 * tol * abs(ref)


def test_complex_diagonal_operator_all_modes_on_the_device():
    rng = np.random.default_rng(2)
    sp, un = ift.RGSpace((8, 4)), ift.UnstructuredDomain(3)
    dom = ift.DomainTuple.make((un, sp))
    full = _cfield(dom, rng)
    partial = _cfield(ift.DomainTuple.make(sp), rng)
    x = _cfield(dom, rng)
    xr = ift.makeField(dom, rng.normal(size=dom.shape))
    for op in (ift.DiagonalOperator(full), ift.DiagonalOperator(partial, dom, 1)):
        for mode in (op.TIMES, op.ADJOINT_TIMES, op.INVERSE_TIMES, op.ADJOINT_INVERSE_TIMES):
            for inp in (x, xr):
                _close(op.apply(inp.at(0), mode), op.apply(inp, mode), 1e-13)
        # adjoint / inverse views and merged chains keep working on the device
        for view in (op.adjoint, op.inverse, op.adjoint.inverse, op.scale(2.0 + 1.0j), op @ op.adjoint):
            _close(view(x.at(0)), view(x), 1e-13)


def test_gaussian_energy_of_complex_residuals_on_the_device():
    """1/2 r^dagger N^-1 r for complex r (energy_operators.py:517-595): value and gradient, device against host."""
    rng = np.random.default_rng(3)
    dom = ift.DomainTuple.make(ift.RGSpace(32))
    data = _cfield(dom, rng)
    icov = ift.DiagonalOperator(ift.makeField(dom, rng.uniform(0.5, 2.0, size=dom.shape)), sampling_dtype=np.complex128)
    x = _cfield(dom, rng)
    for lh in (ift.GaussianEnergy(data=data, inverse_covariance=icov), ift.GaussianEnergy(data=data)):
        vals = []
        for dev in (-1, 0):
            lin = lh(ift.Linearization.make_var(x.at(dev), want_metric=True))
            vals.append((complex(lin.val.asnumpy()[()]), lin.gradient.at(-1), lin.metric(x.at(dev)).at(-1)))
        (v_h, g_h, m_h), (v_d, g_d, m_d) = vals
        assert abs(v_d - v_h) <= 1e-13 * abs(v_h) and abs(v_h.imag) <= 1e-13 * abs(v_h)
        _close(g_d, g_h, 1e-13)
        _close(m_d, m_h, 1e-13)
