"""Operator features added in round 4 (VERDICT r3 "missing" item 6): weighted contractions / IntegrationOperator
(reference contraction_operator.py:44-95, field.py:285-322) and distributors acting on a space that is NOT the last one
(distributors.py:60-104) -- against their numpy definition, on host Fields and (gpu-marked) on device Fields."""
import numpy as np
import pytest
import torch

import nifty_amd as ift

DEVICES = [-1, pytest.param(0, marks=pytest.mark.gpu)]


def _np(f):
    return f.asnumpy()


@pytest.mark.parametrize("dev", DEVICES)
def test_weighted_contraction_and_integration(dev):
    rng = np.random.default_rng(0)
    sp = ift.RGSpace((6, 4), distances=(0.5, 0.25))
    ps = ift.PowerSpace(ift.RGSpace((8, 8)).get_default_codomain())
    un = ift.UnstructuredDomain(3)
    dom = ift.DomainTuple.make((un, sp, ps))
    x = ift.makeField(dom, rng.normal(size=dom.shape)).at(dev)
    xv = _np(x)
    # Field.weight: uniform volume of the RGSpace, per-bin volumes of the PowerSpace
    w = _np(x.weight(1))
    ref = xv * sp.scalar_dvol * np.asarray(ps.dvol)[None, None, None, :]
    assert np.allclose(w, ref, rtol=1e-13, atol=0)
    assert np.allclose(_np(x.weight(-2, spaces=2)), xv * np.asarray(ps.dvol)[None, None, None, :] ** -2.0, rtol=1e-13)
    for spaces, power in (((1,), 1), ((2,), 1), ((1, 2), 2), (None, 1)):
        op = ift.ContractionOperator(dom, spaces, power)
        sl = tuple(range(3)) if spaces is None else spaces
        vol = np.ones(dom.shape)
        if 1 in sl:
            vol = vol * sp.scalar_dvol ** power
        if 2 in sl:
            vol = vol * np.asarray(ps.dvol)[None, None, None, :] ** power
        axes = tuple(a for s in sl for a in dom.axes[s])
        got = _np(op(x))
        assert np.allclose(got, (xv * vol).sum(axis=axes), rtol=1e-12, atol=1e-13)
        y = ift.makeField(op.target, rng.normal(size=op.target.shape)).at(dev)
        back = _np(op.adjoint_times(y))
        shp = [1 if a in axes else n for a, n in enumerate(dom.shape)]
        assert np.allclose(back, np.broadcast_to(_np(y).reshape(shp), dom.shape) * vol, rtol=1e-12)
        # adjointness <y, A x> = <A^T y, x>
        assert abs(op(x).s_vdot(y) - x.s_vdot(op.adjoint_times(y))) < 1e-10 * abs(op(x).s_vdot(y))
    integ = ift.IntegrationOperator(dom, 1)
    assert np.allclose(_np(integ(x)), xv.sum(axis=(1, 2)) * sp.scalar_dvol, rtol=1e-12)
    assert np.allclose(_np(x.sum(2)), xv.sum(axis=3), rtol=1e-12)


@pytest.mark.parametrize("dev", DEVICES)
@pytest.mark.parametrize("space", [0, 1, 2])
def test_distributors_on_any_space(dev, space):
    rng = np.random.default_rng(1)
    hsp = ift.RGSpace((8, 6)).get_default_codomain()
    ps = ift.PowerSpace(hsp)
    pin = np.asarray(ps.pindex)
    others = [ift.UnstructuredDomain(3), ift.RGSpace(5)]
    spaces = others[:]
    spaces.insert(space, hsp)
    tgt = ift.DomainTuple.make(spaces)
    pd = ift.PowerDistributor(tgt, ps, space)
    assert pd.domain[space] is ps and pd.target is tgt
    x = ift.makeField(pd.domain, rng.normal(size=pd.domain.shape)).at(dev)
    y = ift.makeField(tgt, rng.normal(size=tgt.shape)).at(dev)
    ax0 = pd.domain.axes[space][0]
    ref = np.take(_np(x), pin.ravel(), axis=ax0)  # bins -> pixels along the distributed axis
    ref = ref.reshape(_np(x).shape[:ax0] + hsp.shape + _np(x).shape[ax0 + 1:])
    assert np.allclose(_np(pd(x)), ref, rtol=0, atol=0)
    # adjoint: scatter-add of the pixels into their bins
    yv = np.moveaxis(_np(y).reshape(_np(y).shape[:ax0] + (-1,) + _np(y).shape[ax0 + 2:]), ax0, -1)
    bins = np.zeros(yv.shape[:-1] + (ps.shape[0],))
    np.add.at(bins, (..., pin.ravel()), yv)
    assert np.allclose(_np(pd.adjoint_times(y)), np.moveaxis(bins, -1, ax0), rtol=1e-13, atol=1e-13)
    assert abs(pd(x).s_vdot(y) - x.s_vdot(pd.adjoint_times(y))) < 1e-10 * abs(pd(x).s_vdot(y))
    # a plain DOFDistributor with an explicit index field on the same space
    dofdex = ift.makeField(ift.DomainTuple.make(hsp), pin.astype(np.int64))
    dd = ift.DOFDistributor(dofdex, tgt, space)
    xb = ift.makeField(dd.domain, _np(x)).at(dev)
    assert np.allclose(_np(dd(xb)), ref, rtol=0, atol=0)
