"""The nifty.cl-shaped API with device Fields (device_id=0): every operation below runs in libniftyk
kernels.  Same script as tests/test_api_host.py, checked against the reference's golden vectors."""
import numpy as np
import pytest

import nifty_amd as ift
from tests import goldenlib as gl
from tests.test_api_host import CF_ARGS, build

pytestmark = pytest.mark.gpu


def dev(mf):
    return mf.at(0)


@pytest.mark.parametrize("case", gl.MODEL_CASES)
def test_cf_and_hamiltonian_on_device(case):
    z = gl.load("model_" + case)
    m, cfm, cf, lh = build(z, device_id=0)
    x = dev(ift.MultiField.from_raw(cf.domain, gl.latent(z, "x")))
    v = dev(ift.MultiField.from_raw(cf.domain, gl.latent(z, "v")))
    assert x.device_id == 0
    lin = cf(ift.Linearization.make_var(x))
    assert lin.val.device_id == 0
    assert gl.relerr(lin.val.asnumpy(), z["cf"]) < 1e-12
    assert gl.relerr(lin.jac(v).asnumpy(), z["cf_jvp"]) < 1e-11
    w = ift.makeField(cf.target, z["w"], 0)
    assert gl.lat_relerr(lin.jac.adjoint(w).asnumpy(), gl.latent(z, "cf_vjp")) < 1e-11
    ic = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=m["sampling_limit"])
    ham = ift.StandardHamiltonian(lh, ic, prior_sampling_dtype=np.float64)
    hl = ham(ift.Linearization.make_var(x, want_metric=True))
    assert abs(float(hl.val.asnumpy()) - float(z["ham_value"])) < 1e-11 * abs(float(z["ham_value"]))
    assert gl.lat_relerr(hl.gradient.asnumpy(), gl.latent(z, "ham_grad")) < 1e-10
    assert gl.lat_relerr(hl.metric(v).asnumpy(), gl.latent(z, "ham_metric_v")) < 1e-10


@pytest.mark.parametrize("case", ["g1d", "p2d", "p2d_geo"])
def test_sampled_kl_generic_graph_on_device(case):
    z = gl.load("model_" + case)
    m, cfm, cf, lh = build(z, device_id=0)
    x = dev(ift.MultiField.from_raw(cf.domain, gl.latent(z, "x")))
    v = dev(ift.MultiField.from_raw(cf.domain, gl.latent(z, "v")))
    ic = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=m["sampling_limit"])
    ham = ift.StandardHamiltonian(lh, ic, prior_sampling_dtype=np.float64)
    geo = None
    if m["geo"]:
        geo = ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, iteration_limit=3, convergence_level=2), max_cg_iterations=6)
    ift.random.push_sseq_from_seed(m["seed"] + 1)
    try:
        kl = ift.SampledKLEnergy(x, ham, m["n_samples"], geo, mirror_samples=True, device_id=0)
    finally:
        ift.random.pop_sseq()
    tol = 1e-6 if m["geo"] else 1e-9
    for i, s in enumerate(kl.samples.iterator()):
        assert s.device_id == 0
        assert gl.lat_relerr((s - x).asnumpy(), gl.latent(z, f"residual{i}")) < tol, i
    assert abs(kl.value - float(z["kl_value"])) < tol * abs(float(z["kl_value"]))
    assert gl.lat_relerr(kl.gradient.asnumpy(), gl.latent(z, "kl_grad")) < 10 * tol
    assert gl.lat_relerr(kl.apply_metric(v).asnumpy(), gl.latent(z, "kl_metric_v")) < 10 * tol


@pytest.mark.parametrize("fuse", [True, False])
def test_optimize_kl_on_device_matches_reference(fuse):
    """Drop-in check: the reference's optimize_kl result for the same seeds, once through the fused engine
    (fusion pass) and once through the generic operator graph, both on the GPU."""
    z = gl.load("model_g1d")
    m, cfm, cf, lh = build(z, device_id=0)
    ift.random.push_sseq_from_seed(m["seed"] + 2)
    try:
        ic_s = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=m["sampling_limit"])
        mk = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=2),  # noqa: E731
                                    max_cg_iterations=8)
        sl, mean = ift.optimize_kl(lh, 2, m["n_samples"], mk, ic_s, output_directory=None, return_final_position=True,
                                   device_id=0, fuse=fuse)
    finally:
        ift.random.pop_sseq()
    assert mean.device_id == 0
    assert gl.lat_relerr(mean.asnumpy(), gl.latent(z, "okl_mean")) < 1e-5
    for i, s in enumerate(sl.iterator()):
        assert gl.lat_relerr(s.asnumpy(), gl.latent(z, f"okl_sample{i}")) < 1e-5


def test_device_operator_identities():
    sp = ift.RGSpace((16, 32))
    h = ift.HartleyOperator(sp.get_default_codomain(), sp)
    x = ift.from_random(h.domain, device_id=0)
    y = ift.from_random(h.target, device_id=0)
    assert abs(y.s_vdot(h(x)) - h.adjoint(y).s_vdot(x)) < 1e-11
    assert gl.relerr(h.inverse(h(x)).asnumpy(), x.asnumpy()) < 1e-12
    # host and device agree (reference extra.py:519-549)
    assert gl.relerr(h(x).asnumpy(), h(x.at(-1)).asnumpy()) < 1e-12
    pd = ift.PowerDistributor(h.domain)
    a = ift.from_random(pd.domain, device_id=0)
    assert abs(x.s_vdot(pd(a)) - pd.adjoint(x).s_vdot(a)) < 1e-11
    assert gl.relerr(pd.adjoint(x).asnumpy(), pd.adjoint(x.at(-1)).asnumpy()) < 1e-12
    f = ift.FFTOperator(sp)
    xc = ift.from_random(sp, dtype=np.complex128, device_id=0)
    assert gl.relerr(f.inverse(f(xc)).asnumpy(), xc.asnumpy()) < 1e-12
    assert gl.relerr(f(xc).asnumpy(), f(xc.at(-1)).asnumpy()) < 1e-12


@pytest.mark.parametrize("fuse", [True, False])
def test_optimize_kl_geovi_on_device_matches_reference(fuse):
    """geoVI (nonlinear_sampling_minimizer) through optimize_kl: fused engine (FusedGeoEnergy) and generic graph."""
    z = gl.load("model_p2d_geo")
    m, cfm, cf, lh = build(z, device_id=0)
    ift.random.push_sseq_from_seed(m["seed"] + 2)
    try:
        ic_s = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=m["sampling_limit"])
        mk = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=2),  # noqa: E731
                                    max_cg_iterations=8)
        nl = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, iteration_limit=3, convergence_level=2),  # noqa: E731
                                    max_cg_iterations=6)
        sl, mean = ift.optimize_kl(lh, 1, m["n_samples"], mk, ic_s, nonlinear_sampling_minimizer=nl, output_directory=None,
                                   return_final_position=True, device_id=0, fuse=fuse)
    finally:
        ift.random.pop_sseq()
    # ONE iteration (tests/golden/okl1.npz): device sums are order-dependent at 1e-16 (fp64 atomics) and the second
    # iteration of this configuration amplifies that to 1e-3..1e-2 through a discrete decision; the two-iteration
    # vectors are checked on the (deterministic) host path
    z1 = gl.load("okl1")
    lat = lambda pre: {k[len(pre) + 1:]: np.asarray(z1[k]) for k in z1.files if k.startswith(pre + ".")}  # noqa: E731
    assert gl.lat_relerr(mean.asnumpy(), lat("p2d_geo.mean")) < 1e-6
    for i, s in enumerate(sl.iterator()):
        assert gl.lat_relerr(s.asnumpy(), lat(f"p2d_geo.sample{i}")) < 1e-6


def test_device_sampling_rng_option():
    """config sampling_rng="device": the fused engine draws its normal fields on the GPU (seeded per sample from the same
    SeedSequence tree) -- reproducible, mirrored pairs stay exact mirrors, and the result is statistically equivalent to
    (but not bit-identical with) the reference's numpy streams."""
    z = gl.load("model_g1d")
    m, cfm, cf, lh = build(z, device_id=0)

    def run():
        ift.random.push_sseq_from_seed(m["seed"] + 2)
        try:
            ic_s = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=m["sampling_limit"])
            mk = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=2),  # noqa: E731
                                        max_cg_iterations=8)
            return ift.optimize_kl(lh, 1, 1, mk, ic_s, output_directory=None, return_final_position=True, device_id=0)
        finally:
            ift.random.pop_sseq()

    ift.config.update("sampling_rng", "device")
    try:
        sl1, mean1 = run()
        sl2, mean2 = run()
    finally:
        ift.config.update("sampling_rng", "numpy")
    assert gl.lat_relerr(mean1.asnumpy(), mean2.asnumpy()) == 0.0  # reproducible
    s = [x.asnumpy() for x in sl1.iterator()]
    mid = {k: 0.5 * (s[0][k] + s[1][k]) for k in s[0]}
    assert gl.lat_relerr(mid, mean1.asnumpy()) < 1e-12  # mirrored pair
    assert gl.lat_relerr(mean1.asnumpy(), gl.latent(z, "okl_mean")) > 1e-6  # different draws than the numpy stream
    with pytest.raises(ValueError):
        ift.config.update("sampling_rng", "cuda")


def test_large_grid_path_without_host_pindex(monkeypatch):
    """Grids beyond PowerSpace.HOST_PINDEX_LIMIT (1024^3: the 8 N byte host index is never built) must work through the
    user-level API: CorrelatedFieldMaker.finalize + optimize_kl on the device.  The limit is lowered to exercise that
    path on a small grid; the result must equal the ordinary path's."""
    def run():
        ift.PowerSpace._cache.clear()
        ift.random.push_sseq_from_seed(3)
        try:
            sp = ift.RGSpace((64, 64, 128))
            cfm = ift.CorrelatedFieldMaker("")
            cfm.add_fluctuations(sp, CF_ARGS["fluctuations"], CF_ARGS["flexibility"], CF_ARGS["asperity"],
                                 CF_ARGS["loglogavgslope"])
            cfm.set_amplitude_total_offset(2.0, (1e-1, 3e-2))
            cf = cfm.finalize()
            rng = np.random.default_rng(0)
            d = ift.makeField(cf.target, 2.0 + 0.1 * rng.normal(size=sp.shape), 0)
            lh = ift.GaussianEnergy(d, ift.ScalingOperator(cf.target, 100.0, np.float64)) @ cf
            ic = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=4)
            mk = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=1),  # noqa: E731
                                        max_cg_iterations=4)
            return ift.optimize_kl(lh, 1, 1, mk, ic, output_directory=None, return_final_position=True, device_id=0)[1]
        finally:
            ift.random.pop_sseq()

    ref = run().asnumpy()
    monkeypatch.setattr(ift.PowerSpace, "HOST_PINDEX_LIMIT", 1000)
    try:
        got = run().asnumpy()
        with pytest.raises(MemoryError):
            ift.PowerSpace(ift.RGSpace((64, 64, 128)).get_default_codomain()).pindex
    finally:
        ift.PowerSpace._cache.clear()
    # the two paths build the bin index differently but run the same kernels on the same numbers
    assert gl.lat_relerr(got, ref) < 1e-9


def test_optimize_kl_on_grid_the_planner_rejects():
    """RGSpace((11, 26)): 11 and 13 are outside the native radices.  The fused nodes step aside, the generic operator graph
    runs on the device with the chirp-z array seam, and the result equals the host path's (same seeds)."""
    def run(device_id):
        ift.random.push_sseq_from_seed(5)
        try:
            sp = ift.RGSpace((11, 26))
            cfm = ift.CorrelatedFieldMaker("")
            cfm.add_fluctuations(sp, CF_ARGS["fluctuations"], CF_ARGS["flexibility"], CF_ARGS["asperity"],
                                 CF_ARGS["loglogavgslope"])
            cfm.set_amplitude_total_offset(2.0, (1e-1, 3e-2))
            cf = cfm.finalize()
            rng = np.random.default_rng(0)
            d = ift.makeField(cf.target, 2.0 + 0.1 * rng.normal(size=sp.shape), 0)
            lh = ift.GaussianEnergy(d, ift.ScalingOperator(cf.target, 100.0, np.float64)) @ cf
            ic = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=5)
            mk = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=2),  # noqa: E731
                                        max_cg_iterations=5)
            return ift.optimize_kl(lh, 1, 2, mk, ic, output_directory=None, return_final_position=True,
                                   device_id=device_id)[1]
        finally:
            ift.random.pop_sseq()

    on_host, on_dev = run(-1), run(0)
    assert gl.lat_relerr(on_dev.asnumpy(), on_host.asnumpy()) < 1e-8


def test_product_spectrum_on_subspaces_the_planner_rejects():
    """Product of two amplitude spectra over RGSpace((11,)) x RGSpace((5, 7)): every sub-space transform (`space=`) takes the
    chirp-z seam on the device; value, Jacobian and adjoint Jacobian equal the host path's."""
    cfm = ift.CorrelatedFieldMaker("p")
    cfm.add_fluctuations(ift.RGSpace((11,), (0.5,)), (1.0, 5e-1), (1.0, 2e-1), (5e-1, 5e-2), (-3.0, 2e-1), prefix="t")
    cfm.add_fluctuations(ift.RGSpace((5, 7)), (0.7, 3e-1), (1.2, 2e-1), (4e-1, 5e-2), (-2.5, 2e-1), prefix="s")
    cfm.set_amplitude_total_offset(2.0, (1e-1, 3e-2))
    cf = cfm.finalize()
    ift.random.push_sseq_from_seed(21)
    try:
        x = ift.from_random(cf.domain) * 0.3
        v = ift.from_random(cf.domain)
        w = ift.from_random(cf.target)
    finally:
        ift.random.pop_sseq()
    lin_h = cf(ift.Linearization.make_var(x))
    lin_d = cf(ift.Linearization.make_var(x.at(0)))
    assert gl.relerr(lin_d.val.asnumpy(), lin_h.val.asnumpy()) < 1e-12
    assert gl.relerr(lin_d.jac(v.at(0)).asnumpy(), lin_h.jac(v).asnumpy()) < 1e-11
    assert gl.lat_relerr(lin_d.jac.adjoint(w.at(0)).asnumpy(), lin_h.jac.adjoint(w).asnumpy()) < 1e-11
