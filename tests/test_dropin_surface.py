"""The parts of the nifty.cl surface around the hot path that scripts lean on: Field statistics, reverse operator arithmetic,
the full pointwise dictionary, selection / relabelling operators, sugar helpers and the `nifty.cl` import aliases.  Expected
values are plain numpy restatements of the reference's definitions (field.py:419-664, pointwise.py:24-159, operator.py:275-300);
tools/run_reference_tests.py additionally runs the reference's own test files against this package in the build container."""
import subprocess
import sys

import numpy as np
import pytest

import nifty_amd as ift

pmp = pytest.mark.parametrize


def _field(rng, doms, cplx=False):
    dom = ift.DomainTuple.make(doms)
    a = rng.normal(size=dom.shape) + (1j * rng.normal(size=dom.shape) if cplx else 0)
    return ift.makeField(dom, a), a


def test_field_statistics_follow_the_volume_weighted_definitions():
    rng = np.random.default_rng(5)
    doms = [ift.RGSpace((4, 5), (0.3, 0.7)), ift.RGSpace(3), ift.PowerSpace(ift.RGSpace((8,), harmonic=True))]
    f, a = _field(rng, doms)
    w = np.ones_like(a) * 0.3 * 0.7 * (1 / 3) * doms[2].dvol
    vol = w.sum()
    np.testing.assert_allclose(f.s_integrate(), (a * w).sum(), rtol=1e-13)
    np.testing.assert_allclose(f.s_mean(), (a * w).sum() / vol, rtol=1e-13)
    np.testing.assert_allclose(f.s_var(), ((a - (a * w).sum() / vol) ** 2 * w).sum() / vol, rtol=1e-13)
    np.testing.assert_allclose(f.s_std() ** 2, f.s_var(), rtol=1e-13)
    np.testing.assert_allclose(f.total_volume(), vol, rtol=1e-13)
    assert f.scalar_weight() is None and f.scalar_weight((0, 1)) == pytest.approx(0.3 * 0.7 / 3)
    # uniform sub-domains: plain numpy statistics along their axes
    np.testing.assert_allclose(f.mean(0).asnumpy(), a.mean(axis=(0, 1)), rtol=1e-13)
    np.testing.assert_allclose(f.var((0, 1)).asnumpy(), a.var(axis=(0, 1, 2)), rtol=1e-13)
    np.testing.assert_allclose(f.std(1).asnumpy(), a.std(axis=2), rtol=1e-13)
    np.testing.assert_allclose(f.integrate(0).asnumpy(), a.sum(axis=(0, 1)) * 0.21, rtol=1e-13)
    np.testing.assert_allclose(f.prod(1).asnumpy(), a.prod(axis=2), rtol=1e-13)
    np.testing.assert_allclose(f.integrate(2).asnumpy(), (a * doms[2].dvol).sum(axis=3), rtol=1e-13)
    assert f.mean(2).domain is ift.DomainTuple.make(doms[:2])
    g, b = _field(rng, [ift.UnstructuredDomain(3)])
    np.testing.assert_allclose(f.outer(g).asnumpy(), np.multiply.outer(a, b), rtol=1e-15)
    with pytest.raises(TypeError):
        bool(f)


@pmp("name,fn,dfn", [
    ("tan", np.tan, lambda v: 1 / np.cos(v) ** 2), ("sinh", np.sinh, np.cosh), ("cosh", np.cosh, np.sinh),
    ("log10", np.log10, lambda v: 1 / (np.log(10.0) * v)),
    ("sinc", np.sinc, lambda v: np.where(v == 0, 0.0, (np.cos(np.pi * v) - np.sinc(v)) / np.where(v == 0, 1.0, v))),
    ("sign", np.sign, lambda v: np.where(v == 0, np.nan, 0.0)),
    ("unitstep", lambda v: (v >= 0).astype(float), np.zeros_like),
    ("softplus", lambda v: np.where(v > 33, v, np.where(v < -33, 0.0, np.log1p(np.exp(np.clip(v, -33, 33))))),
     lambda v: np.where(v > 33, 1.0, np.where(v < -33, 0.0, 1 / (1 + np.exp(-np.clip(v, -33, 33)))))),
])
def test_remaining_pointwise_functions(name, fn, dfn):
    v = np.array([-40.0, -2.5, -0.3, 0.0, 0.7, 1.9, 35.0])
    if name == "log10":
        v = np.abs(v) + 0.1
    f = ift.makeField(ift.UnstructuredDomain(v.size), v)
    val, der = f.ptw_with_deriv(name)
    np.testing.assert_allclose(val.asnumpy(), fn(v), rtol=1e-14, atol=1e-300)
    np.testing.assert_allclose(der.asnumpy(), dfn(v), rtol=1e-13, atol=1e-300)
    np.testing.assert_array_equal(getattr(ift, name)(f).asnumpy(), val.asnumpy())
    np.testing.assert_array_equal(getattr(f, name)().asnumpy(), val.asnumpy())


def test_exponentiate_and_complex_guards():
    f = ift.makeField(ift.UnstructuredDomain(3), np.array([0.5, 1.0, 2.0]))
    val, der = f.ptw_with_deriv("exponentiate", 1.1)
    np.testing.assert_allclose(val.asnumpy(), 1.1 ** f.asnumpy(), rtol=1e-15)
    np.testing.assert_allclose(der.asnumpy(), np.log(1.1) * 1.1 ** f.asnumpy(), rtol=1e-15)
    z = f + 1j * f
    for name in ("sign", "unitstep"):
        with pytest.raises(TypeError):
            z.ptw(name)
    with pytest.raises(TypeError):
        z.ptw_with_deriv("abs")
    with pytest.raises(TypeError):
        z.clip(0, 1)


def test_reverse_division_and_powers_of_operators():
    dom = ift.RGSpace(6)
    x = ift.from_random(dom).exp()
    y = ift.from_random(dom).exp()
    a, b = ift.FieldAdapter(dom, "a"), ift.FieldAdapter(dom, "b")
    pos = ift.MultiField.from_dict({"a": x, "b": y})
    xv, yv = x.asnumpy(), y.asnumpy()
    cases = [(2.0 / a, 2.0 / xv), (a / y, xv / yv), (y / a, yv / xv), (a ** b, xv ** yv), (3.0 ** a, 3.0 ** xv),
             (a ** y, xv ** yv), (y ** a, yv ** xv), (abs(-a), xv)]
    for op, want in cases:
        np.testing.assert_allclose(op.force(pos).asnumpy(), want, rtol=1e-13)
        ift.extra.check_operator(op, ift.MultiField.from_dict({k: pos[k] for k in op.domain.keys()}), ntries=3, tol=1e-10)
    np.testing.assert_allclose((x ** y).asnumpy(), xv ** yv, rtol=1e-14)
    np.testing.assert_allclose((2.0 ** x).asnumpy(), 2.0 ** xv, rtol=1e-14)
    lin = ift.Linearization.make_var(x)
    np.testing.assert_allclose((lin ** lin).val.asnumpy(), xv ** xv, rtol=1e-14)
    np.testing.assert_allclose((2.0 ** lin).gradient.asnumpy() if False else (2.0 ** lin).val.asnumpy(), 2.0 ** xv, rtol=1e-14)


def test_ducktape_left_and_broadcast_on_fields_and_linearizations():
    dom = ift.RGSpace((4, 2), (0.2, 11.0))
    f = ift.from_random(dom)
    lin = ift.Linearization.make_var(f)
    with pytest.raises(RuntimeError):
        f.ducktape("k")
    with pytest.raises(RuntimeError):
        lin.ducktape("k")
    assert f.ducktape_left("k").domain is ift.MultiDomain.make({"k": dom})
    assert lin.ducktape_left("k").target is ift.MultiDomain.make({"k": dom})
    want = np.broadcast_to(f.asnumpy()[None], (3,) + dom.shape)
    np.testing.assert_array_equal(f.broadcast(0, ift.UnstructuredDomain(3)).asnumpy(), want)
    np.testing.assert_array_equal(lin.broadcast(0, ift.UnstructuredDomain(3)).val.asnumpy(), want)
    assert ift.is_fieldlike(lin) and ift.is_fieldlike(f) and not ift.is_fieldlike(ift.ScalingOperator(dom, 1.0))


def _selection_cases():
    rg, un = ift.RGSpace, ift.UnstructuredDomain
    yield ift.GeometryRemover([rg((3, 4)), rg(5)], 1), False
    yield ift.DomainChangerAndReshaper(rg((3, 4)), un(12)), False
    yield ift.ExtractAtIndices(ift.makeDomain([rg(3), rg((4, 5)), rg(2)]), (np.array([0, 3, 3, 1]), np.array([1, 2, 2, 0])), 1), False
    yield ift.SqueezeOperator([rg((3, 1, 4)), un(1), rg(5)], True), False
    yield ift.SqueezeOperator([rg((3, 1, 4)), un(1), rg(5)]), False
    yield ift.TransposeOperator([rg((3, 2)), un(4), rg(5)], (2, 0, 1)), False
    yield ift.OuterProduct(rg((3, 2)), ift.makeField(un(4), np.arange(4.0) + 1)), False
    yield ift.ValueInserter(rg((3, 2)), (1, 1)), False
    yield ift.DomainTupleFieldInserter(ift.makeDomain([rg(3), rg((4, 5)), un(2)]), 1, (2, 3)), False
    for central in (False, True):
        for shp, new in (((4, 5), (7, 8)), ((5, 4), (6, 9)), ((1, 3), (4, 3))):
            yield ift.FieldZeroPadder(ift.makeDomain([un(2), rg(shp, harmonic=True)]), new, 1, central), False
    for center in (False, True):
        yield ift.SliceOperator([rg((6, 7), (0.1, 0.2)), un(5)], ((3, 4), 2), center), False
    yield ift.ConjugationOperator(rg((3, 4))), True
    yield ift.PartialExtractor(ift.MultiDomain.make({"a": rg(3), "b": un(2)}), ift.MultiDomain.make({"b": un(2)})), False


def test_selection_operators_are_consistent_linear_maps():
    for op, cplx in _selection_cases():
        dt = np.complex128 if cplx else np.float64
        ift.extra.check_linear_operator(op, dt, dt, only_r_linear=cplx)


def test_selection_operators_values():
    rng = np.random.default_rng(2)
    a = rng.normal(size=(2, 4, 5))
    dom = ift.makeDomain([ift.UnstructuredDomain(2), ift.RGSpace((4, 5), harmonic=True)])
    f = ift.makeField(dom, a)
    end = ift.FieldZeroPadder(dom, (6, 5), 1)(f).asnumpy()
    np.testing.assert_array_equal(end[:, :4], a)
    assert not end[:, 4:].any()
    mid = ift.FieldZeroPadder(dom, (7, 5), 1, central=True)
    out = mid(f).asnumpy()
    np.testing.assert_array_equal(out[:, :3], a[:, :3])       # frequencies 0, 1, Nyquist
    np.testing.assert_array_equal(out[:, -2:], a[:, -2:])     # -Nyquist (again), -1
    assert not out[:, 3:5].any()
    back = mid.adjoint_times(mid(f)).asnumpy()
    np.testing.assert_array_equal(back[:, 2], 2 * a[:, 2])    # the Nyquist entry was written twice: its adjoint adds both
    np.testing.assert_array_equal(back[:, [0, 1, 3]], a[:, [0, 1, 3]])
    cut = ift.SliceOperator(dom, (None, (2, 3)), center=True)
    np.testing.assert_array_equal(cut(f).asnumpy(), a[:, 1:3, 1:4])
    assert cut.target[1].distances == dom[1].distances
    pick = ift.ExtractAtIndices(dom, (np.array([0, 3, 3]), np.array([1, 2, 2])), 1)
    np.testing.assert_array_equal(pick(f).asnumpy(), a[:, [0, 3, 3], [1, 2, 2]])
    ones = ift.full(pick.target, 1.0)
    assert pick.adjoint_times(ones).asnumpy()[0, 3, 2] == 2.0
    t = ift.TransposeOperator(dom, (1, 0))
    np.testing.assert_array_equal(t(f).asnumpy(), np.moveaxis(a, 0, 2))
    z = ift.makeField(ift.RGSpace(3), np.array([1 + 2j, 3 - 1j, 0.5j]))
    np.testing.assert_array_equal(ift.Imaginizer(z.domain)(z).asnumpy(), [2.0, -1.0, 0.5])
    np.testing.assert_array_equal(ift.Imaginizer(z.domain).adjoint_times(z.imag).asnumpy(), 1j * z.asnumpy().imag)


def test_power_helpers():
    h = ift.RGSpace((8, 6), (0.5, 0.25), harmonic=True)
    spec = lambda k: 3.0 / (1.0 + k ** 2)  # noqa: E731
    ps = ift.PowerSpace(h)
    op = ift.create_power_operator(h, spec)
    k = h.get_k_length_array().asnumpy()
    # bins hold the mean |k| of their pixels; on this grid every distinct |k| has its own bin
    np.testing.assert_allclose(op(ift.full(h, 1.0)).asnumpy(), spec(ps.k_lengths)[ps.pindex], rtol=1e-14)
    np.testing.assert_allclose(ift.get_signal_variance(spec, h), (spec(ps.k_lengths)[ps.pindex]).sum() * h.scalar_dvol ** 2,
                               rtol=1e-13)
    f = ift.from_random(h, dtype=np.complex128)
    pw = ift.power_analyze(f)
    want = np.bincount(ps.pindex.ravel(), weights=(np.abs(f.asnumpy()) ** 2).ravel()) / ps.rho
    np.testing.assert_allclose(pw.asnumpy(), want, rtol=1e-13)
    both = ift.power_analyze(f, keep_phase_information=True).asnumpy()
    np.testing.assert_allclose(both.real + both.imag, want, rtol=1e-13)
    sm = ift.create_harmonic_smoothing_operator(ift.makeDomain(h), 0, 0.3)
    np.testing.assert_allclose(sm(ift.full(h, 1.0)).asnumpy(), np.exp(-2 * np.pi ** 2 * 0.09 * k ** 2), rtol=1e-14)
    assert ift.get_default_codomain(h) == h.get_default_codomain()
    assert ift.get_default_codomain(ift.makeDomain([ift.UnstructuredDomain(2), h]), 1)[1] == h.get_default_codomain()
    assert k.shape == h.shape


def test_spherical_descriptors_are_geometry_only():
    lm = ift.LMSpace(5, 3)
    assert lm.size == (5 + 1) ** 2 - (5 - 3) * (5 - 3 + 1) and lm.harmonic
    assert list(lm.get_k_length_array().asnumpy()[:8]) == [0, 1, 2, 3, 4, 5, 1, 1]
    np.testing.assert_array_equal(ift.PowerSpace(lm).k_lengths, np.arange(6.0))
    gl = ift.GLSpace(6)
    np.testing.assert_allclose(gl.dvol.sum(), 4 * np.pi, rtol=1e-14)
    assert gl.get_default_codomain() == ift.LMSpace(5, 5) and gl.shape == (6 * 11,)
    hp = ift.HPSpace(4)
    assert hp.size == 192 and hp.total_volume == pytest.approx(4 * np.pi) and hp.get_default_codomain() == ift.LMSpace(8)
    with pytest.raises(TypeError):
        hp.check_codomain(gl)
    f = ift.from_random(hp)
    np.testing.assert_allclose(f.s_integrate(), f.asnumpy().sum() * np.pi / 48, rtol=1e-13)
    with pytest.raises(NotImplementedError):
        ift.HarmonicTransformOperator(ift.LMSpace(8), hp)


def test_correlated_field_with_fixed_or_no_zero_mode():
    dom = ift.RGSpace((8, 6), (0.5, 0.25))
    args = (dom, (1.0, 0.5), (1.0, 0.2), (0.5, 0.05), (-3.0, 0.2))
    for offset_std, keys in ((None, 6), (0, 6), (1, 6), (2.5, 6), ((1.0, 0.3), 7)):
        cfm = ift.CorrelatedFieldMaker("p")
        cfm.add_fluctuations(*args)
        cfm.set_amplitude_total_offset(0.7, offset_std)
        op = cfm.finalize()
        assert len(op.domain.keys()) == keys
        pos = ift.from_random(op.domain)
        ift.extra.check_operator(op, pos, ntries=2, tol=1e-9)
        amp = cfm.amplitude.force(pos).asnumpy()
        if offset_std in (None, 0):
            assert amp[0] == 0.0
            np.testing.assert_allclose(op(pos).s_mean(), 0.7, rtol=1e-12)
    # the simple model carries its amplitude like the reference's
    scf = ift.SimpleCorrelatedField(dom, 0.7, None, *args[1:], prefix="p")
    pos = ift.from_random(scf.domain)
    np.testing.assert_allclose(scf.power_spectrum.force(pos).asnumpy(), scf.amplitude.force(pos).asnumpy() ** 2, rtol=1e-14)


def test_fluctuation_statistics_of_the_maker():
    dom = ift.RGSpace((16, 12), (0.5, 0.25))
    cfm = ift.CorrelatedFieldMaker("")
    cfm.add_fluctuations(dom, (1.3, 1e-3), (1.0, 0.2), (0.5, 0.05), (-4.0, 0.2))
    cfm.set_amplitude_total_offset(1.2, (0.8, 1e-3))
    op = cfm.finalize()
    samples = [op(ift.from_random(op.domain)) for _ in range(60)]
    np.testing.assert_allclose(cfm.total_fluctuation_realized(samples), 1.3, rtol=0.2)
    np.testing.assert_allclose(cfm.offset_amplitude_realized(samples) ** 2, 1.2 ** 2 + 0.8 ** 2, rtol=0.3)
    fluct = cfm.total_fluctuation
    np.testing.assert_allclose(fluct.force(ift.full(op.domain, 0.0)).asnumpy(), 1.3, rtol=1e-5)
    assert cfm.average_fluctuation(0) is cfm.fluctuations[0].fluctuation_amplitude
    with pytest.raises(ValueError):
        cfm.slice_fluctuation(1)


def test_block_diagonal_identity_blocks_and_merging():
    dom = ift.MultiDomain.make({"d1": ift.RGSpace(10), "d2": ift.UnstructuredDomain(2)})
    op = ift.BlockDiagonalOperator(dom, {"d1": ift.ScalingOperator(dom["d1"], 20.0)})
    ift.extra.check_linear_operator(op)
    f = ift.from_random(dom)
    ift.extra.assert_equal(op(f)["d2"], f["d2"])
    for combined, factor in ((op(op), 400.0), (op + op, 40.0)):
        assert type(combined) is ift.BlockDiagonalOperator
        np.testing.assert_allclose(combined(f)["d1"].asnumpy(), factor * f["d1"].asnumpy(), rtol=1e-15)
    np.testing.assert_allclose((op + op)(f)["d2"].asnumpy(), 2 * f["d2"].asnumpy(), rtol=1e-15)


def test_reference_import_paths_resolve_after_compat_install():
    code = ("import nifty_amd.compat as c; c.install(); import nifty.cl as ift; import nifty_amd\n"
            "assert ift is nifty_amd\n"
            "from nifty.cl.minimization.kl_energies import SampledKLEnergyClass\n"
            "from nifty.cl.library.correlated_fields import CorrelatedFieldMaker\n"
            "from nifty.cl.operators.harmonic_operators import HartleyOperator\n"
            "from nifty.cl.utilities import allreduce_sum\n"
            "import nifty.cl.operators.energy_operators as eo\n"
            "assert eo.GaussianEnergy is nifty_amd.GaussianEnergy and allreduce_sum([1.0, 2.0, 4.0], None) == 7.0\n"
            "import numpy as np\n"
            "f = ift.full(ift.RGSpace(4), 2.5)\n"
            "assert isinstance(f.val, ift.AnyArray) and f.val.device_id == -1 and f.val.asnumpy().shape == (4,)\n"
            "assert np.mean(f.val) == 2.5 and np.nansum(f.val.conj() * f.val) == 25.0   # numpy functions on field.val\n"
            "try:\n    from nifty.cl.operators.operator import NoSuchName\n"
            "except ImportError as e:\n    print('missing name reported:', e)\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(__import__("pathlib").Path(__file__).parents[1]))
    assert out.returncode == 0, out.stderr
    assert "missing name reported" in out.stdout


def test_calculate_position_and_exec_time():
    dom = ift.RGSpace(12, harmonic=True)
    op = ift.HarmonicTransformOperator(dom).ptw("exp")
    with ift.random.Context(11):
        truth = 0.1 * ift.from_random(op.domain)
        pos = ift.calculate_position(op, op(truth))
    ift.extra.assert_allclose(op(pos), op(truth), 1e-1, 1e-1)  # the reference's own bar (test_sugar.py:79-89)
    times = ift.exec_time(op, ntries=2)
    assert set(times) == {"apply", "apply_lin", "jac", "jac.adjoint"} and all(t >= 0 for t in times.values())
    lh = ift.GaussianEnergy(op(truth), sampling_dtype=float) @ op
    assert {"gradient", "metric_apply"} <= set(ift.exec_time(lh))


@pytest.mark.gpu
def test_selection_operators_and_statistics_on_the_device():
    for op, cplx in _selection_cases():
        if cplx:
            continue
        ift.extra.check_linear_operator(op, force_device_ids=[0])
    rng = np.random.default_rng(5)
    doms = [ift.RGSpace((4, 5), (0.3, 0.7)), ift.PowerSpace(ift.RGSpace((8,), harmonic=True))]
    f, a = _field(rng, doms)
    d = f.at(0)
    for name in ("s_integrate", "s_mean", "s_var", "s_std"):
        np.testing.assert_allclose(getattr(d, name)(), getattr(f, name)(), rtol=1e-13)
    for name in ("integrate", "mean", "var", "std"):
        for sp in (0, 1):
            np.testing.assert_allclose(getattr(d, name)(sp).asnumpy(), getattr(f, name)(sp).asnumpy(), rtol=1e-13)


@pytest.mark.gpu
@pmp("dtype,tol", [(np.float64, 2e-14), (np.float32, 2e-6)])
def test_every_pointwise_function_has_a_device_kernel(dtype, tol):
    """device value and derivative against the host definitions above, for the whole dictionary of pointwise.py:134-159"""
    rng = np.random.default_rng(9)
    v = np.concatenate([rng.normal(size=4093) * 2.0, [0.0, -40.0, 40.0]]).astype(dtype)
    dom = ift.UnstructuredDomain(v.size)
    cases = [(n, ()) for n in ("tan", "sinc", "sinh", "cosh", "sign", "softplus", "unitstep", "exp", "tanh", "sigmoid", "sin", "cos",
                               "arctan", "expm1", "abs")]
    cases += [("exponentiate", (1.7,)), ("power", (3.0,)), ("clip", (-0.5, 1.25)), ("clip", (None, 0.3)), ("clip", (0.1, None))]
    cases += [(n, ()) for n in ("log", "log10", "log1p", "sqrt", "reciprocal")]
    for name, args in cases:
        x = np.abs(v) + dtype(0.1) if name in ("log", "log10", "log1p", "sqrt", "reciprocal") else v
        if name in ("sinh", "cosh", "exp", "expm1", "tan", "exponentiate", "power"):
            x = np.clip(x, -5, 5)
        host = ift.makeField(dom, x)
        fh, dh = host.ptw_with_deriv(name, *args)
        fd, dd = host.at(0).ptw_with_deriv(name, *args)
        scale_f = np.maximum(np.abs(fh.asnumpy()), 1.0)
        scale_d = np.maximum(np.abs(dh.asnumpy()), 1.0)
        # tan near its poles amplifies the argument's rounding: compare where the function is moderate
        ok = np.abs(fh.asnumpy()) < 50 if name == "tan" else np.ones(v.size, bool)
        np.testing.assert_array_less((np.abs(fd.asnumpy() - fh.asnumpy()) / scale_f)[ok], tol * 8, err_msg=name)
        same_nan = np.isnan(dd.asnumpy()) == np.isnan(dh.asnumpy())
        assert same_nan.all(), name
        fin = ok & ~np.isnan(dh.asnumpy())
        np.testing.assert_array_less((np.abs(dd.asnumpy() - dh.asnumpy()) / scale_d)[fin], tol * 64, err_msg=name + " derivative")
        np.testing.assert_array_equal(host.at(0).ptw(name, *args).asnumpy(), fd.asnumpy())


@pytest.mark.gpu
def test_complex_adjoints_on_the_device():
    dom = ift.RGSpace((6, 5))
    ift.extra.check_linear_operator(ift.VdotOperator(ift.from_random(dom, dtype=np.complex128)), np.complex128, np.complex128,
                                    force_device_ids=[0])
    dofdex = ift.Field.from_raw(dom, np.arange(30).reshape(6, 5) % 4)
    ift.extra.check_linear_operator(ift.DOFDistributor(dofdex), np.complex128, np.complex128, force_device_ids=[0])


def test_sums_of_diagonals_merge_like_the_reference():
    """sum_operator.py:107-140: scalings go into a diagonal, diagonals add up; integer-valued diagonals keep fp64 roots"""
    dom = ift.RGSpace(5)
    op1, op2 = ift.makeOp(ift.Field.full(dom, 2.0)), ift.ScalingOperator(dom, 3.0)
    total = op1 + op2 - (op2 - op1) + op1 + op1 + op2
    assert isinstance(total, ift.DiagonalOperator)
    np.testing.assert_allclose(total(ift.full(dom, 1.0)).asnumpy(), 11.0, rtol=1e-15)
    mixed = op1(op2 + op2)(op1)(op1) - op1(op2)
    assert isinstance(mixed, ift.DiagonalOperator)
    np.testing.assert_allclose(mixed(ift.full(dom, 1.0)).asnumpy(), 42.0, rtol=1e-15)
    whole = ift.makeOp(ift.full(dom, 2))  # int64 values
    x = ift.from_random(dom)
    ift.extra.assert_allclose((whole.get_sqrt().adjoint @ whole.get_sqrt())(x), whole(x), rtol=1e-15)
    with pytest.raises(ValueError):
        ift.makeOp(2.0 + 0j, ift.makeDomain(dom)).get_sqrt()


def test_plot_writes_the_panels_the_path_produces(tmp_path):
    pytest.importorskip("matplotlib")
    import matplotlib

    matplotlib.use("Agg")
    line, image = ift.RGSpace(10), ift.RGSpace((8, 6), distances=1.0)
    power = ift.power_analyze(ift.FFTOperator(image)(ift.from_random(image)))
    history = ift.EnergyHistory()
    for i in range(5):
        history.append((i, (i + 1.0) ** -2))
    p = ift.Plot()
    p.add(ift.from_random(image), title="2d rg", vmin=-1, vmax=1)
    p.add([ift.from_random(line), ift.from_random(line)], title="list 1d rg", label=["1", "2"], alpha=[1, 0.3])
    p.add(power, title="power spectrum")
    p.add(ift.from_random(ift.UnstructuredDomain(10)), title="histogram")
    p.add(history, title="energy")
    p.add(None)
    p.add(ift.from_random(ift.makeDomain({"a": line, "b": line})), title="per key")   # two panels
    p.add(ift.from_random(ift.DomainTuple.make([image, ift.UnstructuredDomain(2)])))   # two image panels
    assert len(p._panels) == 10
    target = tmp_path / "panels.png"
    p.output(title="ten panels", name=str(target), nx=4, ny=3)
    assert target.stat().st_size > 10000
    ift.single_plot(ift.from_random(image), title="one", name=str(tmp_path / "one.png"))
    cf = ift.SimpleCorrelatedField(image, 0.0, (1e-2, 1e-3), (1.0, 0.5), (1.0, 0.5), (0.5, 0.2), (-3.0, 0.5))
    ift.plot_priorsamples(cf, n_samples=2, name=str(tmp_path / "prior.png"))
    assert (tmp_path / "one.png").exists() and (tmp_path / "prior.png").exists()
    with pytest.raises(NotImplementedError):
        q = ift.Plot()
        q.add(ift.from_random(ift.HPSpace(4)))
        q.output(name=str(tmp_path / "sphere.png"))
    with pytest.raises(ValueError):
        ift.Plot().output(name=str(tmp_path / "empty.png"))


@pmp("total_n", [0, 3])
def test_hosted_amplitude_models_equal_the_operator_graph(total_n):
    """the feeder of the fused product node with the amplitude models evaluated on the host (device fields: one packed copy
    each way) against the same graphs evaluated in place: values, JVP, VJP"""
    cfm = ift.CorrelatedFieldMaker("p", total_N=total_n)
    cfm.add_fluctuations(ift.RGSpace((16,), (0.5,)), (1.0, 5e-1), (1.0, 2e-1), (5e-1, 5e-2), (-3.0, 2e-1), prefix="t")
    cfm.add_fluctuations(ift.RGSpace((8, 6)), (0.7, 3e-1), (1.2, 2e-1), (4e-1, 5e-2), (-2.5, 2e-1), prefix="s")
    cfm.set_amplitude_total_offset(2.0, (1e-1, 3e-2))
    cf = cfm.finalize()
    assert cf._hosted is not None and cf._hosted.domain is cf._feeder.domain and cf._hosted.target is cf._feeder.target
    x = ift.from_random(cf.domain) * 0.5
    ift.extra.assert_equal(cf._hosted(x), cf._feeder(x))
    ift.extra.check_operator(cf._hosted, x, ntries=2, tol=1e-9)
    graph, hosted = (op(ift.Linearization.make_var(x)) for op in (cf._feeder, cf._hosted))
    v, w = ift.from_random(cf.domain), ift.from_random(graph.target)
    ift.extra.assert_allclose(hosted.jac(v), graph.jac(v), rtol=1e-13, atol=1e-15)
    ift.extra.assert_allclose(hosted.jac.adjoint(w), graph.jac.adjoint(w), rtol=1e-13, atol=1e-15)


@pytest.mark.gpu
def test_hosted_amplitude_models_on_the_device(monkeypatch):
    """a product-spectrum field on the GPU with the amplitude models on the host (default) and on the device
    (NK_HOSTED_AMPLITUDES=0): value, JVP and VJP of the whole operator through the fused product node"""
    def build():
        cfm = ift.CorrelatedFieldMaker("p")
        cfm.add_fluctuations(ift.RGSpace((64,), (0.5,)), (1.0, 5e-1), (1.0, 2e-1), (5e-1, 5e-2), (-3.0, 2e-1), prefix="t")
        cfm.add_fluctuations(ift.RGSpace((64, 64)), (0.7, 3e-1), (1.2, 2e-1), (4e-1, 5e-2), (-2.5, 2e-1), prefix="s")
        cfm.set_amplitude_total_offset(2.0, (1e-1, 3e-2))
        return cfm.finalize()

    hosted = build()
    monkeypatch.setenv("NK_HOSTED_AMPLITUDES", "0")
    on_device = build()
    assert hosted._fused is not on_device._fused and hosted.domain is on_device.domain
    x = (0.5 * ift.from_random(hosted.domain)).at(0)
    v, w = ift.from_random(hosted.domain).at(0), ift.from_random(hosted.target).at(0)
    a, b = hosted(ift.Linearization.make_var(x)), on_device(ift.Linearization.make_var(x))
    ift.extra.assert_allclose(a.val.at(-1), b.val.at(-1), rtol=1e-12, atol=1e-12)
    ift.extra.assert_allclose(a.jac(v).at(-1), b.jac(v).at(-1), rtol=1e-11, atol=1e-11)
    ga, gb = a.jac.adjoint(w).at(-1), b.jac.adjoint(w).at(-1)
    for key in ga.keys():
        scale = max(np.abs(gb[key].asnumpy()).max(), 1e-300)
        np.testing.assert_array_less(np.abs(ga[key].asnumpy() - gb[key].asnumpy()).max() / scale, 1e-10, err_msg=key)
        assert ga[key].dtype == gb[key].dtype
    assert ga.device_id == -1 and a.jac.adjoint(w).device_id == 0


@pytest.mark.gpu
def test_packed_arithmetic_of_device_multifields_is_exact():
    """element-wise arithmetic on one packed buffer per operand (MultiField.PACK_MAX) gives the bits of the per-key kernels"""
    dom = ift.MultiDomain.make({"a": ift.RGSpace((16, 8)), "b": ift.UnstructuredDomain(5), "c": ift.RGSpace(3),
                                "d": ift.DomainTuple.scalar_domain()})
    for dtype in (np.float64, np.float32):
        x, y = (ift.from_random(dom, dtype=dtype).at(0) for _ in range(2))
        for fn in (lambda u, v: u + v, lambda u, v: u - v, lambda u, v: u * v, lambda u, v: u / v, lambda u, v: 2.5 * u - v * 0.3,
                   lambda u, v: 1.0 / u + (3 - v), lambda u, v: u - 0.25 * v):
            packed = fn(x, y)
            plain = ift.MultiField.from_dict({k: fn(x[k], y[k]) for k in dom.keys()})
            assert packed._flat is not None and packed["a"].dtype == np.dtype(dtype)
            ift.extra.assert_equal(packed.at(-1), plain.at(-1))
        assert x.s_vdot(y) == sum(x[k].s_vdot(y[k]) for k in dom.keys())
        import pickle

        back = pickle.loads(pickle.dumps((x + y).at(-1)))
        ift.extra.assert_equal(back, (x + y).at(-1))
    mixed = ift.MultiField.from_dict({"a": ift.from_random(dom["a"]).at(0), "b": ift.from_random(dom["b"], dtype=np.float32).at(0),
                                      "c": ift.from_random(dom["c"]).at(0), "d": ift.from_random(dom["d"]).at(0)})
    assert (mixed + mixed)._flat is None  # mixed dtypes: per key
    ift.extra.assert_equal((mixed + mixed).at(-1), (2.0 * mixed).at(-1))
