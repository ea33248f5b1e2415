"""The reference's operator-consistency harness (extra.py check_linear_operator / check_operator; SURVEY 4: the bulk
of the reference's operator tests are calls of these two functions) applied to the operators of the path.
Without a GPU the checks run on the host Fields; with the gpu marker the same calls also cover device 0 and the
host/device agreement."""
import numpy as np
import pytest

import nifty_amd as ift
from nifty_amd.extra import check_linear_operator, check_operator


def _linear_operators():
    sp1, sp2, sp3 = ift.RGSpace((32,), (0.1,)), ift.RGSpace((16, 12), (0.3, 0.2)), ift.RGSpace((8, 6, 10))
    rng = np.random.default_rng(2)
    ops = []
    for sp in (sp1, sp2, sp3):
        h = sp.get_default_codomain()
        ops += [(ift.HartleyOperator(sp), np.float64, np.float64), (ift.FFTOperator(sp), np.complex128, np.complex128),
                (ift.HarmonicTransformOperator(h), np.float64, np.float64),
                (ift.PowerDistributor(h), np.float64, np.float64),
                (ift.DiagonalOperator(ift.makeField(sp, rng.uniform(0.5, 2.0, sp.shape))), np.float64, np.float64),
                (ift.ScalingOperator(sp, 2.5), np.float64, np.float64),
                (ift.HarmonicSmoothingOperator(sp, 0.02), np.float64, np.float64),
                (ift.ContractionOperator(sp, None), np.float64, np.float64)]
    # operators on a sub-space of a DomainTuple (`space=`, reference harmonic_operators.py:59-75, contraction_operator.py)
    dt = ift.DomainTuple.make((sp1, sp2))
    for space in (0, 1):
        ops += [(ift.HartleyOperator(dt, space=space), np.float64, np.float64),
                (ift.FFTOperator(dt, space=space), np.complex128, np.complex128),
                (ift.ContractionOperator(dt, space), np.float64, np.float64)]
    ops.append((ift.HarmonicTransformOperator(ift.DomainTuple.make((sp1.get_default_codomain(), sp2)), space=0),
                np.float64, np.float64))
    nlos = 9
    ops.append((ift.LOSResponse(sp2, rng.uniform(0, 4, (2, nlos)), rng.uniform(0, 2.4, (2, nlos))), np.float64, np.float64))
    ops.append((ift.MaskOperator(ift.makeField(sp2, rng.uniform(size=sp2.shape) < 0.4)), np.float64, np.float64))
    return ops


def _run_linear(force):
    ift.random.push_sseq_from_seed(5)
    try:
        for op, dd, td in _linear_operators():
            check_linear_operator(op, dd, td, atol=1e-11, rtol=1e-11, force_device_ids=force)
    finally:
        ift.random.pop_sseq()


def _model(sp, kind):
    cfm = ift.CorrelatedFieldMaker("")
    cfm.add_fluctuations(sp, (1.0, 5e-1), (1.0, 2e-1), (5e-1, 5e-2), (-3.0, 2e-1))
    cfm.set_amplitude_total_offset(2.0, (1e-1, 3e-2))
    cf = cfm.finalize()
    if kind == "cf":
        return cf
    if kind == "exp":
        return cf.ptw("exp")
    rng = np.random.default_rng(3)
    if kind == "gauss":
        d = ift.makeField(cf.target, rng.normal(size=sp.shape))
        return ift.GaussianEnergy(d, ift.ScalingOperator(cf.target, 4.0, np.float64)) @ cf
    d = ift.makeField(cf.target, rng.poisson(5.0, size=sp.shape).astype(np.int64))
    return ift.PoissonianEnergy(d) @ cf.ptw("exp")


def _run_nonlinear(force):
    ift.random.push_sseq_from_seed(6)
    try:
        for sp in (ift.RGSpace((32,)), ift.RGSpace((12, 8))):
            for kind in ("cf", "exp", "gauss", "poisson"):
                op = _model(sp, kind)
                check_operator(op, 0.3 * ift.from_random(op.domain), tol=1e-8, ntries=2, force_device_ids=force)
    finally:
        ift.random.pop_sseq()


def test_linear_operators_consistent_host():
    _run_linear([-1])


def test_nonlinear_operators_consistent_host():
    _run_nonlinear([-1])


@pytest.mark.gpu
def test_linear_operators_consistent_device():
    _run_linear([0])


@pytest.mark.gpu
def test_nonlinear_operators_consistent_device():
    _run_nonlinear([0])


def test_check_functions_reject_wrong_types():
    sp = ift.RGSpace((8,))
    with pytest.raises(TypeError):
        check_linear_operator(ift.HartleyOperator(sp).ptw("exp"))
    with pytest.raises(TypeError):
        check_operator(3, None)
