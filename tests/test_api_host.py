"""Host (device_id = -1) tests of the nifty.cl-shaped API against the golden vectors generated from the
reference: the same user-level script (CorrelatedFieldMaker -> GaussianEnergy/PoissonianEnergy ->
StandardHamiltonian -> SampledKLEnergy -> NewtonCG -> optimize_kl) with ``import nifty_amd as ift``.
This is config 1 of BASELINE.json ("plumbing, no GPU") and exercises the generic operator graph."""
import numpy as np
import pytest

import nifty_amd as ift
from tests import goldenlib as gl

CF_ARGS = dict(fluctuations=(1.0, 5e-1), flexibility=(1.0, 2e-1), asperity=(5e-1, 5e-2), loglogavgslope=(-3.0, 2e-1))


def build(z, device_id=-1):
    m = gl.meta(z)
    sp = ift.RGSpace(m["shape"], m["distances"])
    cfm = ift.CorrelatedFieldMaker("")
    cfm.add_fluctuations(sp, CF_ARGS["fluctuations"], CF_ARGS["flexibility"], CF_ARGS["asperity"], CF_ARGS["loglogavgslope"])
    cfm.set_amplitude_total_offset(2.0, (1e-1, 3e-2))
    cf = cfm.finalize()
    sig = cf if m["nonlin"] is None else cf.ptw(m["nonlin"])
    if m["kind"] == "gaussian":
        d = ift.makeField(sig.target, z["data"], device_id)
        ic = z["icov"]
        if ic.shape == ():
            N_inv = ift.ScalingOperator(sig.target, float(ic), np.float64)
        else:
            N_inv = ift.makeOp(ift.makeField(sig.target, ic, device_id), sampling_dtype=np.float64)
        lh = ift.GaussianEnergy(d, N_inv) @ sig
    else:
        lh = ift.PoissonianEnergy(ift.makeField(sig.target, z["data"], device_id)) @ sig
    return m, cfm, cf, lh


@pytest.mark.parametrize("case", gl.MODEL_CASES)
def test_cf_and_hamiltonian(case):
    z = gl.load("model_" + case)
    m, cfm, cf, lh = build(z)
    x = ift.MultiField.from_raw(cf.domain, gl.latent(z, "x"))
    v = ift.MultiField.from_raw(cf.domain, gl.latent(z, "v"))
    lin = cf(ift.Linearization.make_var(x))
    assert gl.relerr(lin.val.asnumpy(), z["cf"]) < 1e-12
    assert gl.relerr(lin.jac(v).asnumpy(), z["cf_jvp"]) < 1e-11
    w = ift.makeField(cf.target, z["w"])
    assert gl.lat_relerr(lin.jac.adjoint(w).asnumpy(), gl.latent(z, "cf_vjp")) < 1e-11
    amp = cfm.amplitude
    assert gl.relerr(amp.force(x).asnumpy(), z["amplitude"]) < 1e-12
    ic = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=m["sampling_limit"])
    ham = ift.StandardHamiltonian(lh, ic, prior_sampling_dtype=np.float64)
    hl = ham(ift.Linearization.make_var(x, want_metric=True))
    assert abs(float(hl.val.asnumpy()) - float(z["ham_value"])) < 1e-11 * abs(float(z["ham_value"]))
    assert gl.lat_relerr(hl.gradient.asnumpy(), gl.latent(z, "ham_grad")) < 1e-10
    assert gl.lat_relerr(hl.metric(v).asnumpy(), gl.latent(z, "ham_metric_v")) < 1e-10


@pytest.mark.parametrize("case", ["g1d", "p2d", "g2d_dist", "p2d_geo"])
def test_sampled_kl_and_newton(case):
    z = gl.load("model_" + case)
    m, cfm, cf, lh = build(z)
    x = ift.MultiField.from_raw(cf.domain, gl.latent(z, "x"))
    v = ift.MultiField.from_raw(cf.domain, gl.latent(z, "v"))
    ic = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=m["sampling_limit"])
    ham = ift.StandardHamiltonian(lh, ic, prior_sampling_dtype=np.float64)
    geo = None
    if m["geo"]:
        geo = ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, iteration_limit=3, convergence_level=2), max_cg_iterations=6)
    ift.random.push_sseq_from_seed(m["seed"] + 1)
    try:
        kl = ift.SampledKLEnergy(x, ham, m["n_samples"], geo, mirror_samples=True)
    finally:
        ift.random.pop_sseq()
    tol = 1e-6 if m["geo"] else 1e-9
    for i, s in enumerate(kl.samples.iterator()):
        assert gl.lat_relerr((s - x).asnumpy(), gl.latent(z, f"residual{i}")) < tol, i
    assert abs(kl.value - float(z["kl_value"])) < tol * abs(float(z["kl_value"]))
    assert gl.lat_relerr(kl.gradient.asnumpy(), gl.latent(z, "kl_grad")) < 10 * tol
    assert gl.lat_relerr(kl.apply_metric(v).asnumpy(), gl.latent(z, "kl_metric_v")) < 10 * tol
    mini = ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=2), max_cg_iterations=8)
    kl2, _ = mini(kl)
    assert abs(kl2.value - float(z["kl_min_value"])) < 100 * tol * abs(float(z["kl_min_value"]))


def test_optimize_kl_host_matches_reference():
    z = gl.load("model_g1d")
    m, cfm, cf, lh = build(z)
    ift.random.push_sseq_from_seed(m["seed"] + 2)
    try:
        ic_s = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=m["sampling_limit"])
        mk = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=2),  # noqa: E731
                                    max_cg_iterations=8)
        sl, mean = ift.optimize_kl(lh, 2, m["n_samples"], mk, ic_s, output_directory=None, return_final_position=True)
    finally:
        ift.random.pop_sseq()
    assert gl.lat_relerr(mean.asnumpy(), gl.latent(z, "okl_mean")) < 1e-6
    for i, s in enumerate(sl.iterator()):
        assert gl.lat_relerr(s.asnumpy(), gl.latent(z, f"okl_sample{i}")) < 1e-5


def _resume_case(tmp_path, device_id, fuse=True, comm=None, rank=0):
    """Interrupted + resumed run equals the uninterrupted one (reference test_mpi/test_optimize_kl.py:117-146): the
    sample files, the mean, last_finished_iteration and the RNG state file on disk carry the whole state."""
    z = gl.load("model_g1d")
    m, cfm, cf, lh = build(z, device_id)
    ic_s = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=3)
    mk = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, iteration_limit=1), max_cg_iterations=4)  # noqa: E731
    kw = dict(return_final_position=True, device_id=device_id, fuse=fuse, comm=comm)

    def run(total, outdir, resume, **extra):
        ift.random.push_sseq_from_seed(5)
        try:
            return ift.optimize_kl(lh, total, 1, mk, ic_s, output_directory=str(outdir), resume=resume, **kw, **extra)
        finally:
            ift.random.pop_sseq()

    sl_full, full = run(3, tmp_path / "a", False)
    with pytest.raises(Exception):
        run(3, tmp_path / "b", False,
            terminate_callback=lambda i: (_ for _ in ()).throw(RuntimeError("stop")) if i == 1 else False)
    sl_res, resumed = run(3, tmp_path / "b", True)
    assert resumed.device_id == device_id
    assert gl.lat_relerr(resumed.asnumpy(), full.asnumpy()) < 1e-12
    for a, b in zip(sl_res.local_iterator(), sl_full.local_iterator()):
        assert gl.lat_relerr(a.asnumpy(), b.asnumpy()) < 1e-12
    return full


def test_optimize_kl_resume(tmp_path):
    _resume_case(tmp_path, -1)


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", [True, False])
def test_optimize_kl_resume_on_device(tmp_path, fuse):
    """The same with device fields (samples are written as host pickles and come back to the device), on the fused engine
    and on the generic graph."""
    _resume_case(tmp_path, 0, fuse=fuse)


def test_linear_operator_modes_and_errors():
    sp = ift.RGSpace((8, 8))
    h = ift.HartleyOperator(sp.get_default_codomain(), sp)
    x = ift.from_random(h.domain)
    y = ift.from_random(h.target)
    # adjointness <y, A x> == <A^T y, x>  (reference extra.py:220-231)
    assert abs(y.s_vdot(h(x)) - h.adjoint(y).s_vdot(x)) < 1e-12
    # A A^-1 = 1 (extra.py:234-244)
    assert gl.relerr(h.inverse(h(x)).asnumpy(), x.asnumpy()) < 1e-12
    with pytest.raises(ValueError):
        h(y)  # wrong domain -> ValueError like utilities.check_object_identity
    with pytest.raises(NotImplementedError):
        ift.HarmonicTransformOperator(h.domain, sp).apply(x, ift.LinearOperator.INVERSE_TIMES)
    with pytest.raises(NotImplementedError):
        h.apply(x, 3)
    d = ift.makeOp(ift.from_random(h.domain).exp())
    assert gl.relerr(d.inverse(d(x)).asnumpy(), x.asnumpy()) < 1e-12
    ch = h @ d
    assert abs(y.s_vdot(ch(x)) - ch.adjoint(y).s_vdot(x)) < 1e-12
    pd = ift.PowerDistributor(h.domain)
    a = ift.from_random(pd.domain)
    assert abs(x.s_vdot(pd(a)) - pd.adjoint(x).s_vdot(a)) < 1e-12


def test_fft_roundtrip_host():
    # reference test_fft_operator.py:38-56
    sp = ift.RGSpace((16, 8), distances=(0.3, 0.7))
    op = ift.FFTOperator(sp)
    x = ift.from_random(sp, dtype=np.complex128)
    assert gl.relerr(op.inverse(op(x)).asnumpy(), x.asnumpy()) < 1e-12


def test_fusion_pass_matches_graph():
    from nifty_amd.optimize_kl import match_fused

    z = gl.load("model_p2d")
    m, cfm, cf, lh = build(z)
    kw = match_fused(lh)
    assert kw is not None and kw["likelihood"] == "poisson" and kw["nonlin"] == "exp" and kw["shape"] == (32, 32)
    z = gl.load("model_g1d")
    m, cfm, cf, lh = build(z)
    kw = match_fused(lh)
    assert kw["likelihood"] == "gaussian" and kw["icov"] == pytest.approx(100.0) and kw["nonlin"] is None
    # BASELINE config 4: Mask @ LOSResponse @ sigmoid(cf) becomes ONE sparse matrix of the fused engine (the kept rows)
    zl = gl.load("los")
    sp = ift.RGSpace((16, 16))
    cfm = ift.CorrelatedFieldMaker("")
    cfm.add_fluctuations(sp, CF_ARGS["fluctuations"], CF_ARGS["flexibility"], CF_ARGS["asperity"], CF_ARGS["loglogavgslope"])
    cfm.set_amplitude_total_offset(2.0, (1e-1, 3e-2))
    cf = cfm.finalize()
    R = ift.LOSResponse(sp, zl["c4.starts"], zl["c4.ends"])
    Mk = ift.MaskOperator(ift.makeField(R.target, zl["c4.flags"]))
    for resp, n_data in ((Mk @ R @ cf.ptw("sigmoid"), Mk.target.size), (R @ cf.ptw("sigmoid"), R.target.size), (R @ cf, R.target.size)):
        d = ift.from_random(resp.target)
        kw = match_fused(ift.GaussianEnergy(d, ift.ScalingOperator(resp.target, 1e3, np.float64)) @ resp)
        assert kw is not None and kw["response"].n_data == n_data and kw["response"].n_pix == 256
        x = np.random.default_rng(0).normal(size=256)
        want = (Mk(R(ift.makeField(sp, x.reshape(16, 16)))) if n_data == Mk.target.size else R(ift.makeField(sp, x.reshape(16, 16)))).asnumpy()
        assert gl.relerr(kw["response"].host_matrix @ x, want) < 1e-12
    # a response the fusion pass does not know stays on the generic graph
    other = ift.ScalingOperator(R.target, 2.0) @ R @ cf
    assert match_fused(ift.GaussianEnergy(ift.from_random(other.target), ift.ScalingOperator(other.target, 1e3, np.float64)) @ other) is None


def _smooth_numpy(x, dist, sigma):
    """Gaussian smoothing as a Fourier multiplier exp(-2 pi^2 sigma^2 |k|^2) (harmonic_operators.py:340-380,
    rg_space.py:116-128, 164-171)."""
    k2 = 0.0
    for ax, (n, d) in enumerate(zip(x.shape, dist)):
        k = np.minimum(np.arange(n), n - np.arange(n)) / (n * d)
        k2 = k2 + (k ** 2).reshape([-1 if i == ax else 1 for i in range(x.ndim)])
    return np.fft.ifftn(np.fft.fftn(x) * np.exp(-2.0 * np.pi ** 2 * sigma ** 2 * k2)).real


@pytest.mark.parametrize("device_id", [-1, pytest.param(0, marks=pytest.mark.gpu)])
def test_harmonic_smoothing_and_counting_operator(device_id):
    rng = np.random.default_rng(0)
    for shape, dist in [((32,), (0.1,)), ((16, 24), (0.3, 0.2))]:
        sp = ift.RGSpace(shape, dist)
        x = rng.normal(size=shape)
        op = ift.HarmonicSmoothingOperator(sp, 0.25)
        got = op(ift.makeField(sp, x, device_id)).asnumpy()
        assert gl.relerr(got, _smooth_numpy(x, dist, 0.25)) < 1e-12
    # on ONE space of a DomainTuple (harmonic_operators.py:340-380 with `space`): the other spaces are carried along
    un = ift.UnstructuredDomain(3)
    for doms, space, lead in (((un, sp), 1, True), ((sp, un), 0, False)):
        op = ift.HarmonicSmoothingOperator(doms, 0.25, space=space)
        xs = rng.normal(size=op.domain.shape)
        got = op(ift.makeField(op.domain, xs, device_id)).asnumpy()
        for i in range(3):
            ref = _smooth_numpy(xs[i] if lead else xs[..., i], dist, 0.25)
            assert gl.relerr(got[i] if lead else got[..., i], ref) < 1e-12
    with pytest.raises(ValueError):
        ift.HarmonicSmoothingOperator((un, sp), 0.25)  # space must be given for several sub-domains
    assert isinstance(ift.HarmonicSmoothingOperator(sp, 0.0), ift.ScalingOperator)
    with pytest.raises(ValueError):
        ift.HarmonicSmoothingOperator(sp, -1.0)
    with pytest.raises(TypeError):
        ift.HarmonicSmoothingOperator(sp.get_default_codomain(), 1.0)
    c = ift.CountingOperator(sp)
    f = ift.makeField(sp, x, device_id)
    assert c(f) is f
    lin = c(ift.Linearization.make_var(f))
    lin.jac(f)
    lin.jac.adjoint(f)
    lin.jac.adjoint(f)
    assert (c.count_apply, c.count_apply_lin, c.count_jac, c.count_jac_adj) == (1, 1, 1, 2)
    assert "Adjoint Jacobian" in c.report()


@pytest.mark.parametrize("device_id", [-1, pytest.param(0, marks=pytest.mark.gpu)])
def test_lbfgs_and_steepest_descent_match_reference(device_id):
    """descent_minimizers.py:138-468 on the Hamiltonian of the g1d model (tests/golden/minimizers.npz)."""
    z, zm = gl.load("model_g1d"), gl.load("minimizers")
    m, cfm, cf, lh = build(z, device_id)
    x = ift.MultiField.from_raw(cf.domain, gl.latent(z, "x"), device_id)
    ham = ift.StandardHamiltonian(lh, ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=10),
                                  prior_sampling_dtype=np.float64)
    runs = {"lbfgs": ift.L_BFGS(ift.AbsDeltaEnergyController(0.1, iteration_limit=6)),
            "steepest": ift.SteepestDescent(ift.GradientNormController(iteration_limit=3)),
            # vector-free L-BFGS (:264-468) and the stochastic controller (iteration_controllers.py:426-497)
            "vlbfgs": ift.VL_BFGS(ift.AbsDeltaEnergyController(0.1, iteration_limit=6), max_history_length=3),
            "lbfgs_stoch": ift.L_BFGS(ift.StochasticAbsDeltaEnergyController(5.0, iteration_limit=8, memory_length=3))}
    for name, mini in runs.items():
        e, _ = mini(ift.EnergyAdapter(x, ham, want_metric=True))
        assert abs(e.value - float(zm[f"{name}.value"])) < 1e-9 * abs(float(zm[f"{name}.value"]))
        assert gl.lat_relerr(e.position.asnumpy(), gl.latent(zm, f"{name}.pos")) < 1e-8


def test_relaxed_newton_on_quadratic():
    """-M^{-1} g is exact for a quadratic with an invertible metric: one step reaches the minimum
    (descent_minimizers.py:149-163)."""
    sp = ift.RGSpace((16,))
    rng = np.random.default_rng(1)
    A = ift.DiagonalOperator(ift.makeField(sp, rng.uniform(0.5, 2.0, 16)))
    b = ift.makeField(sp, rng.normal(size=16))
    e = ift.QuadraticEnergy(ift.full(sp, 0.0), A, b)
    e, status = ift.RelaxedNewton(ift.GradientNormController(iteration_limit=3, tol_abs_gradnorm=1e-10))(e)
    assert gl.relerr(e.position.asnumpy(), (A.inverse_times(b)).asnumpy()) < 1e-12


def _lat(z, prefix):
    return {k[len(prefix) + 1:]: np.asarray(z[k]) for k in z.files if k.startswith(prefix + ".")}


@pytest.mark.parametrize("device_id", [-1, pytest.param(0, marks=pytest.mark.gpu)])
def test_constants_and_point_estimates_match_reference(device_id):
    """SampledKLEnergy / optimize_kl with `constants` and `point_estimates` (kl_energies.py:162-297,
    optimize_kl.py:408-421): value, gradient on the variable keys, metric, samples, a NewtonCG step and a 2-iteration
    optimize_kl run against tests/golden/constants.npz."""
    z, zc = gl.load("model_g1d"), gl.load("constants")
    m, cfm, cf, lh = build(z, device_id)
    x = ift.MultiField.from_raw(cf.domain, gl.latent(z, "x"), device_id)
    ic = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=6)
    ham = ift.StandardHamiltonian(lh, ic, prior_sampling_dtype=np.float64)
    cst, pes = ["fluctuations", "zeromode"], ["loglogavgslope", "zeromode"]
    ift.random.push_sseq_from_seed(43)
    try:
        kl = ift.SampledKLEnergy(x, ham, 2, None, mirror_samples=True, constants=cst, point_estimates=pes,
                                 device_id=device_id)
    finally:
        ift.random.pop_sseq()
    assert set(kl.position.keys()) == set(cf.domain.keys()) - set(cst)
    assert abs(kl.value - float(zc["kl.value"])) < 1e-9 * abs(float(zc["kl.value"]))
    assert gl.lat_relerr(kl.gradient.asnumpy(), _lat(zc, "kl.grad")) < 1e-8
    v = ift.MultiField.from_raw(kl.position.domain, _lat(zc, "kl.v"), device_id)
    assert gl.lat_relerr(kl.apply_metric(v).asnumpy(), _lat(zc, "kl.metric_v")) < 1e-8
    for i, s in enumerate(kl.samples.iterator()):
        ref = _lat(zc, f"kl.sample{i}")
        assert set(s.keys()) == set(ref) and gl.lat_relerr(s.asnumpy(), ref) < 1e-8
    mini = ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=2), max_cg_iterations=8)
    kl2, _ = mini(kl)
    assert abs(kl2.value - float(zc["kl.min_value"])) < 1e-7 * abs(float(zc["kl.min_value"]))
    assert gl.lat_relerr(kl2.position.asnumpy(), _lat(zc, "kl.min_pos")) < 1e-6
    ift.random.push_sseq_from_seed(44)
    try:
        mk = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=2),  # noqa: E731
                                    max_cg_iterations=8)
        sl, mean = ift.optimize_kl(lh, 2, 1, mk, ic, constants=["fluctuations"], point_estimates=["flexibility"],
                                   output_directory=None, return_final_position=True, initial_position=x,
                                   device_id=device_id)
    finally:
        ift.random.pop_sseq()
    assert gl.lat_relerr(mean.asnumpy(), _lat(zc, "okl.mean")) < 1e-5
    assert np.array_equal(mean["fluctuations"].asnumpy(), x["fluctuations"].asnumpy())
    for i, s in enumerate(sl.iterator()):
        assert gl.lat_relerr(s.asnumpy(), _lat(zc, f"okl.sample{i}")) < 1e-5


def test_one_iteration_geovi_goldens_on_host():
    """The one-iteration geoVI vectors the device tests use (tests/golden/okl1.npz) hold on the host path too."""
    z, z1 = gl.load("model_p2d_geo"), gl.load("okl1")
    m, cfm, cf, lh = build(z)
    ift.random.push_sseq_from_seed(m["seed"] + 2)
    try:
        ic_s = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=m["sampling_limit"])
        mk = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=2),  # noqa: E731
                                    max_cg_iterations=8)
        nl = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, iteration_limit=3, convergence_level=2),  # noqa: E731
                                    max_cg_iterations=6)
        sl, mean = ift.optimize_kl(lh, 1, m["n_samples"], mk, ic_s, nonlinear_sampling_minimizer=nl, output_directory=None,
                                   return_final_position=True)
    finally:
        ift.random.pop_sseq()
    assert gl.lat_relerr(mean.asnumpy(), _lat(z1, "p2d_geo.mean")) < 1e-6


@pytest.mark.parametrize("device_id", [-1, pytest.param(0, marks=pytest.mark.gpu)])
def test_minisanity_table(device_id):
    """extra.minisanity (extra.py:552-723) on the g1d model with its golden MGVI residuals: the table text below is the
    reference's own output for these inputs; device Fields are reduced by nk_stats."""
    z = gl.load("model_g1d")
    m, cfm, cf, lh = build(z, device_id)
    x = ift.MultiField.from_raw(cf.domain, gl.latent(z, "x"), device_id)
    res = [ift.MultiField.from_raw(cf.domain, gl.latent(z, f"residual{i}"), device_id) for i in range(2)]
    sl = ift.ResidualSampleList(x, res, [False, False])
    table, vals = ift.extra.minisanity(lh, sl, terminal_colors=False, return_values=True)
    expect = [("<None>", "68.6 ± 3.2", "0.0 ± 0.2", "128"), ("asperity", "0.0 ± 0.0", "0.2 ± 0.1", "1"),
              ("flexibility", "4.1 ± 2.7", "-0.5 ± 2.8", "1"), ("spectrum", "1.2 ± 0.2", "-0.0 ± 0.1", "126"),
              ("xi", "1.1 ± 0.1", "-0.0 ± 0.1", "128"), ("zeromode", "0.3 ± 0.1", "-0.1 ± 0.8", "1")]
    lines = {ln.split()[0]: ln for ln in table.splitlines() if ln.startswith("  ")}
    for key, chi, mean, ndof in expect:
        assert chi in lines[key] and mean in lines[key] and lines[key].split()[-2] == ndof, lines[key]
    assert vals["ndof"]["latent_variables"]["xi"] == 128 and vals["nigndof"]["data_residuals"]["<None>"] == 0
    with pytest.raises(TypeError):
        ift.extra.minisanity(lh, [x])


class _MemoryH5Group(dict):
    """In-memory stand-in for an h5py group / file (the image has no h5py): records what save_to_hdf5 writes."""

    def __init__(self):
        super().__init__()
        self.attrs, self.closed = {}, False

    def create_group(self, name):
        assert name not in self
        self[name] = _MemoryH5Group()
        return self[name]

    def create_dataset(self, name, data=None):
        assert name not in self
        self[name] = np.array(data)

    def close(self):
        self.closed = True


def test_sample_list_hdf5_export(tmp_path, monkeypatch):
    """SampleListBase.save_to_hdf5 (sample_list.py:104-184): file layout (samples/0.., stats/mean, stats/standard deviation,
    MultiFields as sub-groups, domain attribute), the reference's argument checks, and ImportError without h5py."""
    import sys
    import types

    z = gl.load("model_g1d")
    m, cfm, cf, lh = build(z, -1)
    x = ift.MultiField.from_raw(cf.domain, gl.latent(z, "x"), -1)
    res = [ift.MultiField.from_raw(cf.domain, gl.latent(z, f"residual{i}"), -1) for i in range(2)]
    sl = ift.ResidualSampleList(x, res, [False, False])
    name = str(tmp_path / "samples.h5")
    if "h5py" not in sys.modules:
        try:
            import h5py  # noqa: F401
        except ImportError:
            with pytest.raises(ImportError):
                sl.save_to_hdf5(name, samples=True)
    files = {}
    fake = types.ModuleType("h5py")
    fake.File = lambda fn, mode: files.setdefault(fn, _MemoryH5Group())
    monkeypatch.setitem(sys.modules, "h5py", fake)
    with pytest.raises(ValueError):
        sl.save_to_hdf5(name)
    sl.save_to_hdf5(name, samples=True, mean=True, std=True)
    f = files[name]
    assert f.closed and f.attrs["nifty domain"] == repr(sl.domain)
    assert sorted(f) == ["samples", "stats"] and sorted(f["samples"]) == ["0", "1"]
    assert sorted(f["stats"]) == ["mean", "standard deviation"]
    samples = list(sl.iterator())
    mean, var = sl.sample_stat()
    for key in cf.domain.keys():
        np.testing.assert_array_equal(f["samples"]["1"][key], samples[1][key].asnumpy())
        np.testing.assert_allclose(f["stats"]["mean"][key], mean[key].asnumpy(), rtol=1e-15)
        np.testing.assert_allclose(f["stats"]["standard deviation"][key], np.sqrt(var[key].asnumpy()), rtol=1e-15)
    # through an operator: Field entries are data sets, the operator is described in the attributes
    files.clear()
    sl.save_to_hdf5(name, op=cf, mean=True)
    f = files[name]
    assert sorted(f) == ["stats"] and list(f["stats"]) == ["mean"]
    np.testing.assert_allclose(f["stats"]["mean"], sl.average(cf).asnumpy(), rtol=1e-15)
    assert f.attrs["nifty domain"] == repr(cf.target) and f.attrs["nifty operator domain"] == repr(cf.domain)
    # an existing file is only replaced on request
    open(name, "w").close()
    with pytest.raises(RuntimeError):
        sl.save_to_hdf5(name, samples=True)
    sl.save_to_hdf5(name, samples=True, overwrite=True)


def _product_cf():
    cfm = ift.CorrelatedFieldMaker("p")
    cfm.add_fluctuations(ift.RGSpace((16,), (0.5,)), (1.0, 5e-1), (1.0, 2e-1), (5e-1, 5e-2), (-3.0, 2e-1), prefix="t")
    cfm.add_fluctuations(ift.RGSpace((8, 6)), (0.7, 3e-1), (1.2, 2e-1), (4e-1, 5e-2), (-2.5, 2e-1), prefix="s")
    cfm.set_amplitude_total_offset(2.0, (1e-1, 3e-2))
    return cfm, cfm.finalize()


@pytest.mark.parametrize("device_id", [-1, pytest.param(0, marks=pytest.mark.gpu)])
def test_product_spectrum_correlated_field(device_id):
    """Two add_fluctuations calls: harmonic transforms over sub-spaces (`space=`), partial contraction / broadcast, product
    of the distributed amplitudes (correlated_fields.py:713-764) against tests/golden/product_cf.npz, including a
    Hamiltonian and one MGVI iteration on the generic operator graph (on the device with nk_cumsum for the log-log
    integrations)."""
    z = gl.load("product_cf")
    cfm, cf = _product_cf()
    assert cf.target.shape == (16, 8, 6)
    with pytest.raises(NotImplementedError):
        cfm.amplitude
    x = ift.MultiField.from_raw(cf.domain, _lat(z, "x"), device_id)
    v = ift.MultiField.from_raw(cf.domain, _lat(z, "v"), device_id)
    w = ift.makeField(cf.target, z["w"], device_id)
    lin = cf(ift.Linearization.make_var(x))
    assert gl.relerr(lin.val.asnumpy(), z["cf"]) < 1e-12
    assert gl.relerr(lin.jac(v).asnumpy(), z["cf_jvp"]) < 1e-11
    assert gl.lat_relerr(lin.jac.adjoint(w).asnumpy(), _lat(z, "cf_vjp")) < 1e-11
    d = ift.makeField(cf.target, z["data"], device_id)
    lh = ift.GaussianEnergy(d, ift.ScalingOperator(cf.target, 100.0, np.float64)) @ cf
    ic = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=6)
    ham = ift.StandardHamiltonian(lh, ic, prior_sampling_dtype=np.float64)
    hl = ham(ift.Linearization.make_var(x, want_metric=True))
    assert abs(float(hl.val.asnumpy()) - float(z["ham_value"])) < 1e-11 * abs(float(z["ham_value"]))
    assert gl.lat_relerr(hl.gradient.asnumpy(), _lat(z, "ham_grad")) < 1e-10
    assert gl.lat_relerr(hl.metric(v).asnumpy(), _lat(z, "ham_metric_v")) < 1e-10
    ift.random.push_sseq_from_seed(9)
    try:
        mk = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=2),  # noqa: E731
                                    max_cg_iterations=8)
        sl, mean = ift.optimize_kl(lh, 1, 1, mk, ic, output_directory=None, return_final_position=True,
                                   initial_position=x, device_id=device_id)
    finally:
        ift.random.pop_sseq()
    assert gl.lat_relerr(mean.asnumpy(), _lat(z, "okl_mean")) < 1e-6
    # device fields run through ONE fused node (nk_product_field / nk_hartley_fused / nk_product_marginal), host fields
    # through the operator graph
    calls = cf.fused_node.calls
    assert (min(calls.values()) > 0) if device_id >= 0 else (max(calls.values()) == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["octant", "matern", "three", "f32", "copies"])
def test_fused_product_field_equals_the_operator_graph(case):
    """`_ProductFieldNode` against the generic operator graph of the same maker on the device: a power-of-two grid (the
    register-resident pipeline: OCTANT amplitude fields and octant sums), a Matern amplitude times a standard one, three
    sub-spaces, and fp32 fields."""
    cfm = ift.CorrelatedFieldMaker("q", total_N=2 if case == "copies" else 0)
    dtype, tol = np.float64, 1e-12
    if case == "copies":  # two field copies with their own spectra on a power-of-two grid: one octant field per copy
        cfm.add_fluctuations(ift.RGSpace((64, 64), (0.5, 0.25)), (1.0, 0.5), (1.2, 0.4), (0.4, 0.2), (-3.0, 0.5), prefix="s",
                             dofdex=[0, 1])
        cfm.add_fluctuations(ift.RGSpace((64,)), (0.8, 0.3), (1.0, 0.3), None, (-2.0, 0.4), prefix="e", dofdex=[0, 0])
    elif case == "octant" or case == "f32":
        cfm.add_fluctuations(ift.RGSpace((64, 64), (0.5, 0.25)), (1.0, 0.5), (1.2, 0.4), (0.4, 0.2), (-3.0, 0.5), prefix="s")
        cfm.add_fluctuations(ift.RGSpace((64,)), (0.8, 0.3), (1.0, 0.3), None, (-2.0, 0.4), prefix="e")
        if case == "f32":
            dtype, tol = np.float32, 2e-5
    elif case == "matern":
        cfm.add_fluctuations_matern(ift.RGSpace((12, 10), (0.5, 0.25)), (1.0, 0.3), (2.0, 0.5), (-4.0, 0.5), prefix="m")
        cfm.add_fluctuations(ift.RGSpace((8,)), (0.8, 0.3), (1.0, 0.3), (0.4, 0.2), (-2.0, 0.4), prefix="e")
    else:
        for n, name in ((8, "a"), (6, "b"), (10, "c")):
            cfm.add_fluctuations(ift.RGSpace((n,)), (0.8, 0.3), (1.0, 0.3), (0.4, 0.2), (-2.5, 0.4), prefix=name)
    cfm.set_amplitude_total_offset(1.5, (1e-1, 3e-2), dofdex=[0, 1] if case == "copies" else None)
    cf = cfm.finalize()
    node = cf.fused_node
    ift.random.push_sseq_from_seed(31)
    try:
        x = ift.from_random(cf.domain, dtype=dtype, device_id=0) * 0.5
        v = ift.from_random(cf.domain, dtype=dtype, device_id=0)
        w = ift.from_random(cf.target, dtype=dtype, device_id=0)
    finally:
        ift.random.pop_sseq()
    fused = cf(ift.Linearization.make_var(x))
    assert min(node.calls["value"], 1) == 1
    generic = cf._generic(ift.Linearization.make_var(x))
    if case in ("octant", "f32", "copies"):
        assert node._setup(x["qxi"].val.dtype, x["qxi"].val.device)["octant"]
    assert gl.relerr(fused.val.asnumpy(), generic.val.asnumpy()) < tol
    assert gl.relerr(fused.jac(v).asnumpy(), generic.jac(v).asnumpy()) < 10 * tol
    assert gl.lat_relerr(fused.jac.adjoint(w).asnumpy(), generic.jac.adjoint(w).asnumpy()) < 10 * tol
    assert node.calls["times"] == 1 and node.calls["adjoint"] == 1


@pytest.mark.parametrize("device_id", [-1, pytest.param(0, marks=pytest.mark.gpu)])
def test_matern_amplitude_correlated_field(device_id):
    """add_fluctuations_matern (correlated_fields.py:231-275, 577-657): value, JVP and VJP against the reference."""
    z = gl.load("product_cf")
    cfm = ift.CorrelatedFieldMaker("m")
    cfm.add_fluctuations_matern(ift.RGSpace((12, 10), (0.5, 0.25)), (1.0, 0.3), (2.0, 0.5), (-4.0, 0.5), prefix="a")
    cfm.set_amplitude_total_offset(1.0, (1e-1, 3e-2))
    cf = cfm.finalize()
    assert set(cf.domain.keys()) == {"macutoff", "maloglogslope", "mascale", "mxi", "mzeromode"}
    x = ift.MultiField.from_raw(cf.domain, _lat(z, "matern.x"), device_id)
    v = ift.MultiField.from_raw(cf.domain, _lat(z, "matern.v"), device_id)
    w = ift.makeField(cf.target, z["matern.w"], device_id)
    lin = cf(ift.Linearization.make_var(x))
    assert gl.relerr(lin.val.asnumpy(), z["matern.cf"]) < 1e-12
    assert gl.relerr(lin.jac(v).asnumpy(), z["matern.cf_jvp"]) < 1e-11
    assert gl.lat_relerr(lin.jac.adjoint(w).asnumpy(), _lat(z, "matern.cf_vjp")) < 1e-11


@pytest.mark.parametrize("device_id", [-1, pytest.param(0, marks=pytest.mark.gpu)])
def test_studentt_and_bernoulli_energies(device_id):
    """StudentTEnergy / BernoulliEnergy (energy_operators.py:704-792) against tests/golden/likelihoods.npz, plus the
    Jacobian-vs-finite-differences harness."""
    from nifty_amd.extra import check_operator

    z = gl.load("likelihoods")
    sp = ift.RGSpace(z["x"].shape)
    x, v = ift.makeField(sp, z["x"], device_id), ift.makeField(sp, z["v"], device_id)
    energies = {"bernoulli": ift.BernoulliEnergy(ift.makeField(sp, z["d"])), "studentt": ift.StudentTEnergy(sp, 3.0),
                "studentt_field": ift.StudentTEnergy(sp, ift.makeField(sp, z["theta"])),
                # InverseGammaEnergy (energy_operators.py:643-701), scalar and field alpha
                "invgamma": ift.InverseGammaEnergy(ift.makeField(sp, z["beta"])),
                "invgamma_field": ift.InverseGammaEnergy(ift.makeField(sp, z["beta"]), ift.makeField(sp, z["theta"]))}
    for name, e in energies.items():
        lin = e(ift.Linearization.make_var(x, want_metric=True))
        assert abs(float(lin.val.asnumpy()) - float(z[f"{name}.value"])) < 1e-12 * abs(float(z[f"{name}.value"]))
        assert gl.relerr(lin.gradient.asnumpy(), z[f"{name}.grad"]) < 1e-12
        assert gl.relerr(lin.metric(v).asnumpy(), z[f"{name}.metric_v"]) < 1e-12
        assert gl.relerr(e.get_transformation()[1](x).asnumpy(), z[f"{name}.trafo"]) < 1e-12
    if device_id < 0:
        ift.random.push_sseq_from_seed(2)
        try:
            check_operator(energies["studentt"], ift.makeField(sp, z["x"]), tol=1e-8, ntries=2)
            check_operator(energies["bernoulli"], ift.makeField(sp, z["x"]), tol=1e-7, ntries=2)
            check_operator(energies["invgamma_field"], ift.makeField(sp, z["x"]), tol=1e-7, ntries=2)
        finally:
            ift.random.pop_sseq()
    with pytest.raises(ValueError):
        ift.BernoulliEnergy(ift.makeField(sp, np.full(sp.shape, 2, dtype=np.int64)))
    with pytest.raises(TypeError):
        ift.BernoulliEnergy(ift.makeField(sp, z["x"]))
    with pytest.raises(TypeError):
        ift.InverseGammaEnergy(z["beta"])
    with pytest.raises(TypeError):
        ift.InverseGammaEnergy(ift.makeField(sp, z["beta"].astype(np.float32)))


@pytest.mark.parametrize("tag", ["noasp", "both"])
@pytest.mark.parametrize("device_id", [-1, pytest.param(0, marks=pytest.mark.gpu)])
def test_reduced_amplitude_models(tag, device_id):
    """add_fluctuations with asperity=None (integrated Wiener process without the asperity term) and with flexibility =
    asperity = None (pure power law), correlated_fields.py:351-363, against tests/golden/reduced_amp.npz; flexibility
    may not be disabled on its own."""
    z = gl.load("reduced_amp")
    cfm = ift.CorrelatedFieldMaker("r")
    cfm.add_fluctuations(ift.RGSpace((12, 10), (0.5, 0.25)), (1.0, 5e-1), (1.2, 2e-1) if tag != "both" else None, None,
                         (-3.0, 2e-1), prefix="a")
    cfm.set_amplitude_total_offset(1.5, (1e-1, 3e-2))
    cf = cfm.finalize()
    assert sorted(cf.domain.keys()) == sorted(_lat(z, f"{tag}.x").keys())
    x = ift.MultiField.from_raw(cf.domain, _lat(z, f"{tag}.x"), device_id)
    v = ift.MultiField.from_raw(cf.domain, _lat(z, f"{tag}.v"), device_id)
    w = ift.makeField(cf.target, z[f"{tag}.w"], device_id)
    lin = cf(ift.Linearization.make_var(x))
    assert gl.relerr(lin.val.asnumpy(), z[f"{tag}.cf"]) < 1e-12
    assert gl.relerr(lin.jac(v).asnumpy(), z[f"{tag}.cf_jvp"]) < 1e-11
    assert gl.lat_relerr(lin.jac.adjoint(w).asnumpy(), _lat(z, f"{tag}.cf_vjp")) < 1e-11
    with pytest.raises(ValueError):
        ift.CorrelatedFieldMaker("").add_fluctuations(ift.RGSpace((8,)), (1.0, 0.5), None, (0.5, 0.05), (-3.0, 0.2))


@pytest.mark.parametrize("device_id", [-1, pytest.param(0, marks=pytest.mark.gpu)])
def test_napprox_preconditioner_matches_reference(device_id):
    """The sampled diagonal preconditioner `napprox` (kl_energies.py:127-128, descent_minimizers.py:201-203,
    probing.py:24-75, 142-152): MGVI samples and a NewtonCG run against tests/golden/napprox.npz (RNG order included:
    the probing draws come from the current stream before the sample seeds are spawned)."""
    z, zn = gl.load("model_g1d"), gl.load("napprox")
    m, cfm, cf, lh = build(z, device_id)
    x = ift.MultiField.from_raw(cf.domain, gl.latent(z, "x"), device_id)
    ham = ift.StandardHamiltonian(lh, ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=4),
                                  prior_sampling_dtype=np.float64)
    ift.random.push_sseq_from_seed(44)
    try:
        kl = ift.SampledKLEnergy(x, ham, 2, None, mirror_samples=True, napprox=3, device_id=device_id)
    finally:
        ift.random.pop_sseq()
    assert abs(kl.value - float(zn["kl.value"])) < 1e-8 * abs(float(zn["kl.value"]))
    for i, s in enumerate(kl.samples.iterator()):
        assert gl.lat_relerr(s.asnumpy(), _lat(zn, f"kl.sample{i}")) < 1e-7
    ift.random.push_sseq_from_seed(45)
    try:
        mini = ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, iteration_limit=2), napprox=3, max_cg_iterations=4)
        e, _ = mini(ift.EnergyAdapter(x, ham, want_metric=True))
    finally:
        ift.random.pop_sseq()
    assert abs(e.value - float(zn["newton.value"])) < 1e-8 * abs(float(zn["newton.value"]))
    assert gl.lat_relerr(e.position.asnumpy(), _lat(zn, "newton.pos")) < 1e-7
    # StatCalculator: running mean / unbiased variance
    sc = ift.StatCalculator()
    vals = [ift.full(cf.target, float(v)) for v in (1.0, 2.0, 4.0)]
    for v in vals:
        sc.add(v)
    assert abs(float(sc.mean.asnumpy().ravel()[0]) - 7.0 / 3.0) < 1e-14
    assert abs(float(sc.var.asnumpy().ravel()[0]) - np.var([1.0, 2.0, 4.0], ddof=1)) < 1e-14


@pytest.mark.parametrize("device_id", [-1, pytest.param(0, marks=pytest.mark.gpu)])
@pytest.mark.parametrize("full", [True, False])
def test_variable_covariance_gaussian_energy(device_id, full):
    """VariableCovarianceGaussianEnergy (energy_operators.py:355-450) on a MultiDomain {residual, inverse covariance}:
    value, gradient, metric (exact Fisher / local transformation) and get_transformation against
    tests/golden/likelihoods.npz."""
    z = gl.load("likelihoods")
    sp = ift.RGSpace(z["vcg_r"].shape)
    e = ift.VariableCovarianceGaussianEnergy(sp, "res", "icov", np.float64, use_full_fisher=full)
    pos = ift.MultiField.from_raw(e.domain, {"res": z["vcg_r"], "icov": z["vcg_i"]}, device_id)
    vv = ift.MultiField.from_raw(e.domain, {"res": z["vcg_vr"], "icov": z["vcg_vi"]}, device_id)
    lin = e(ift.Linearization.make_var(pos, want_metric=True))
    tag = "vcg_full" if full else "vcg_local"
    assert abs(float(lin.val.asnumpy()) - float(z[f"{tag}.value"])) < 1e-12 * abs(float(z[f"{tag}.value"]))
    assert gl.lat_relerr(lin.gradient.asnumpy(), _lat(z, f"{tag}.grad")) < 1e-12
    assert gl.lat_relerr(lin.metric(vv).asnumpy(), _lat(z, f"{tag}.metric_v")) < 1e-12
    assert gl.lat_relerr(e.get_transformation()[1](pos).asnumpy(), _lat(z, f"{tag}.trafo")) < 1e-12
    with pytest.raises(NotImplementedError):
        ift.VariableCovarianceGaussianEnergy(sp, "res", "icov", np.complex128)


@pytest.mark.parametrize("device_id", [-1, pytest.param(0, marks=pytest.mark.gpu)])
def test_categorical_and_averaged_energies(device_id):
    """CategoricalEnergy (energy_operators.py:795-850) and AveragedEnergy (:934-971) against tests/golden/likelihoods.npz."""
    z = gl.load("likelihoods")
    csp = ift.RGSpace(z["cat_d"].shape)
    e = ift.CategoricalEnergy(ift.makeField(csp, z["cat_d"]))
    x, v = ift.makeField(csp, z["cat_x"], device_id), ift.makeField(csp, z["cat_v"], device_id)
    lin = e(ift.Linearization.make_var(x, want_metric=True))
    assert abs(float(lin.val.asnumpy()) - float(z["categorical.value"])) < 1e-12 * abs(float(z["categorical.value"]))
    assert gl.relerr(lin.gradient.asnumpy(), z["categorical.grad"]) < 1e-12
    assert gl.relerr(lin.metric(v).asnumpy(), z["categorical.metric_v"]) < 1e-12
    assert gl.relerr(e.get_transformation()[1](x).asnumpy(), z["categorical.trafo"]) < 1e-12
    with pytest.raises(ValueError):
        ift.CategoricalEnergy(ift.makeField(csp, np.ones(csp.shape, dtype=np.int64)))
    sp = ift.RGSpace(z["x"].shape)
    e = ift.AveragedEnergy(ift.StudentTEnergy(sp, 3.0), [ift.makeField(sp, r) for r in z["avg_samples"]])
    lin = e(ift.Linearization.make_var(ift.makeField(sp, z["x"], device_id), want_metric=True))
    assert abs(float(lin.val.asnumpy()) - float(z["averaged.value"])) < 1e-12 * abs(float(z["averaged.value"]))
    assert gl.relerr(lin.gradient.asnumpy(), z["averaged.grad"]) < 1e-12
    assert gl.relerr(lin.metric(ift.makeField(sp, z["v"], device_id)).asnumpy(), z["averaged.metric_v"]) < 1e-12


@pytest.mark.parametrize("device_id", [-1, pytest.param(0, marks=pytest.mark.gpu)])
def test_likelihood_sum_matches_reference(device_id):
    """lh1 + lh2 (energy_operators.py:211-303: several data sets on one model; disjoint union of the data spaces keyed by
    the likelihood names): Hamiltonian value / gradient / metric, normalised residual, MGVI and geoVI samples against
    tests/golden/lhsum.npz."""
    z, zs = gl.load("model_g1d"), gl.load("lhsum")
    m, cfm, cf, lh1 = build(z, device_id)
    lh2 = ift.PoissonianEnergy(ift.makeField(cf.target, zs["counts"], device_id)) @ cf.exp()
    lh2.name = "counts"
    lh = lh1 + lh2
    assert type(lh).__name__ == "_LikelihoodSum" and lh._all_names() == ["Likelihood 0", "counts"]
    with pytest.raises(RuntimeError):
        lh.name = "x"
    x = ift.MultiField.from_raw(cf.domain, gl.latent(z, "x"), device_id)
    v = ift.MultiField.from_raw(cf.domain, _lat(zs, "v"), device_id)
    ic = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=6)
    ham = ift.StandardHamiltonian(lh, ic, prior_sampling_dtype=np.float64)
    lin = ham(ift.Linearization.make_var(x, want_metric=True))
    assert abs(float(lin.val.asnumpy()) - float(zs["ham.value"])) < 1e-11 * abs(float(zs["ham.value"]))
    assert gl.lat_relerr(lin.gradient.asnumpy(), _lat(zs, "ham.grad")) < 1e-10
    assert gl.lat_relerr(lin.metric(v).asnumpy(), _lat(zs, "ham.metric_v")) < 1e-10
    nres = lh.normalized_residual(x)
    assert set(nres.keys()) == {"Likelihood 0", "counts"}
    assert gl.lat_relerr(nres.asnumpy(), _lat(zs, "nres")) < 1e-10
    for name, geo in (("mgvi", None), ("geovi", ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, iteration_limit=2), max_cg_iterations=4))):
        ift.random.push_sseq_from_seed(21)
        try:
            kl = ift.SampledKLEnergy(x, ham, 1, geo, mirror_samples=True, device_id=device_id)
        finally:
            ift.random.pop_sseq()
        assert abs(kl.value - float(zs[f"{name}.kl_value"])) < 1e-7 * abs(float(zs[f"{name}.kl_value"]))
        for i, smp in enumerate(kl.samples.iterator()):
            assert gl.lat_relerr(smp.asnumpy(), _lat(zs, f"{name}.sample{i}")) < 1e-6


@pytest.mark.parametrize("device_id", [-1, pytest.param(0, marks=pytest.mark.gpu)])
def test_total_N_correlated_fields_with_dofdex(device_id):
    """CorrelatedFieldMaker(total_N=3): three fields on a leading UnstructuredDomain, a 2-D spectrum shared by two of them
    (dofdex [0, 0, 1]) times a 1-D spectrum with one model per field, two zero-mode models
    (correlated_fields.py:211-231, 277-386, 435-764) against tests/golden/totaln_cf.npz."""
    z = gl.load("totaln_cf")
    cfm = ift.CorrelatedFieldMaker("t", total_N=3)
    cfm.add_fluctuations(ift.RGSpace((8, 6), (0.5, 0.25)), (1.0, 0.5), (1.2, 0.4), (0.4, 0.2), (-3.0, 0.5), prefix="sp",
                         dofdex=[0, 0, 1])
    cfm.add_fluctuations(ift.RGSpace((10,)), (0.8, 0.3), (1.0, 0.3), None, (-2.0, 0.4), prefix="en", dofdex=[0, 1, 2])
    cfm.set_amplitude_total_offset(0.5, (1e-1, 3e-2), dofdex=[0, 1, 1])
    cf = cfm.finalize()
    assert cf.target.shape == (3, 8, 6, 10)
    assert cf.domain["tspspectrum"].shape[0] == 2 and cf.domain["tenspectrum"].shape[0] == 3
    assert cf.domain["tzeromode"].shape == (2,)
    x = ift.MultiField.from_raw(cf.domain, _lat(z, "x"), device_id)
    v = ift.MultiField.from_raw(cf.domain, _lat(z, "v"), device_id)
    w = ift.makeField(cf.target, z["w"], device_id)
    lin = cf(ift.Linearization.make_var(x))
    assert gl.relerr(lin.val.asnumpy(), z["cf"]) < 1e-12
    assert gl.relerr(lin.jac(v).asnumpy(), z["cf_jvp"]) < 1e-11
    assert gl.lat_relerr(lin.jac.adjoint(w).asnumpy(), _lat(z, "cf_vjp")) < 1e-11
    for i, na in enumerate(cfm.get_normalized_amplitudes()):
        assert gl.relerr(na.force(x).asnumpy(), z[f"namp{i}"]) < 1e-12
    d = ift.makeField(cf.target, z["data"], device_id)
    lh = ift.GaussianEnergy(d, ift.ScalingOperator(cf.target, 100.0, np.float64)) @ cf
    ham = ift.StandardHamiltonian(lh, ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=6),
                                  prior_sampling_dtype=np.float64)
    hl = ham(ift.Linearization.make_var(x, want_metric=True))
    assert abs(float(hl.val.asnumpy()) - float(z["ham_value"])) < 1e-11 * abs(float(z["ham_value"]))
    assert gl.lat_relerr(hl.gradient.asnumpy(), _lat(z, "ham_grad")) < 1e-10
    assert gl.lat_relerr(hl.metric(v).asnumpy(), _lat(z, "ham_metric_v")) < 1e-10
    calls = cf.fused_node.calls  # three field copies through the fused node on the device, none on the host
    assert (min(calls.values()) > 0) if device_id >= 0 else (max(calls.values()) == 0)
    with pytest.raises(ValueError):
        cfm.add_fluctuations(ift.RGSpace((4,)), (1.0, 0.5), (1.2, 0.4), (0.4, 0.2), (-3.0, 0.5), dofdex=[0, 1])
    with pytest.raises(NotImplementedError):
        cfm.add_fluctuations_matern(ift.RGSpace((4,)), (1.0, 0.3), (2.0, 0.5), (-4.0, 0.5))


def test_controller_announces_a_forced_stop():
    """_LevelController.stops_at_next_check: True exactly when the next check() hits the iteration limit -- what lets the
    in-place CG skip a residual refresh whose result nobody would read (minimization._inplace_steps)."""
    from nifty_amd.minimization import CONTINUE, CONVERGED, _ScalarEnergyView, _forced_stop

    ctrl = ift.AbsDeltaEnergyController(deltaE=1e-30, iteration_limit=3)
    e = _ScalarEnergyView(1.0, 1.0)
    assert ctrl.start(e) == CONTINUE and not ctrl.stops_at_next_check()
    assert ctrl.check(_ScalarEnergyView(0.5, 1.0)) == CONTINUE and not _forced_stop(ctrl)
    assert ctrl.check(_ScalarEnergyView(0.25, 1.0)) == CONTINUE and _forced_stop(ctrl)
    assert ctrl.check(_ScalarEnergyView(0.125, 1.0)) == CONVERGED
    assert not _forced_stop(ift.GradientNormController(tol_abs_gradnorm=1e-3))  # no iteration limit: never forced
    assert not _forced_stop(object())


def test_top_level_helpers_of_the_reference():
    """Small names of nifty.cl's top level that user scripts lean on (operators/operator.py:653-673, utilities.py:516-520,
    ducc_dispatch.py:31-46, logger.py:21-33): same meaning here, where Field / Linearization are not Operator subclasses."""
    import logging

    sp = ift.RGSpace(8)
    op, f = ift.ScalingOperator(sp, 2.0), ift.Field.full(sp, 1.0)
    lh, lin = ift.GaussianEnergy(data=f), ift.Linearization.make_var(f)
    assert ift.is_operator(op) and ift.is_operator(lh) and not ift.is_operator(f) and not ift.is_operator(lin)
    assert ift.is_linearization(lin) and not ift.is_linearization(op) and not ift.is_linearization(f)
    assert ift.is_likelihood_energy(lh) and not ift.is_likelihood_energy(op) and not ift.is_likelihood_energy(f)
    assert ift.is_fieldlike(f) and not ift.is_fieldlike(op)
    ift.set_nthreads(3)
    assert ift.nthreads() == 3
    ift.set_nthreads(1)
    with pytest.raises(AssertionError):
        ift.myassert(0)
    ift.myassert(1)
    assert isinstance(ift.logger, logging.Logger) and issubclass(ift.ResidualSampleList, ift.SampleListBase)
