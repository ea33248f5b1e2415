"""LOSResponse + MaskOperator (BASELINE config 4) against vectors generated from the real reference
(tests/golden/make_golden.py::los_cases): the sparse matrices themselves, TIMES / ADJOINT_TIMES, and a config-4 style
model -- sigmoid(correlated field) -> LOS -> mask -> Gaussian noise with geoVI -- on the host and on the GPU."""
import numpy as np
import pytest

import nifty_amd as ift
from oracle import nifty_oracle as orc
from tests import goldenlib as gl
from tests.test_api_host import CF_ARGS

CASES = ("a", "b", "c", "d")


def _case(z, tag):
    sig = z[f"{tag}.sigmas"]
    return (tuple(int(i) for i in z[f"{tag}.shape"]), tuple(float(d) for d in z[f"{tag}.dist"]), z[f"{tag}.starts"],
            z[f"{tag}.ends"], None if np.isnan(sig).any() else sig)


@pytest.mark.parametrize("tag", CASES)
def test_oracle_los_matrix_equals_reference(tag):
    z = gl.load("los")
    shape, dist, starts, ends, sig = _case(z, tag)
    dense = orc.los_dense(shape, dist, starts, ends, sig)
    assert np.max(np.abs(dense - z[f"{tag}.dense"])) <= 1e-7 * np.max(np.abs(z[f"{tag}.dense"]))
    assert gl.relerr(dense.astype(np.float64) @ z[f"{tag}.x"].reshape(-1), z[f"{tag}.times"]) < 1e-12


def test_oracle_mask_equals_reference():
    z = gl.load("los")
    assert np.array_equal(orc.mask_apply(z["mask.flags"], z["mask.x"]), z["mask.times"])
    assert np.array_equal(orc.mask_apply(z["mask.flags"], z["mask.y"], adjoint=True), z["mask.adjoint"])


def _check_los(tag, device_id):
    from scipy.sparse import csr_matrix

    z = gl.load("los")
    shape, dist, starts, ends, sig = _case(z, tag)
    sp = ift.RGSpace(shape, dist)
    R = ift.LOSResponse(sp, starts, ends, sig)
    assert R.target.shape == (starts.shape[1],)
    dense = csr_matrix((R._wgt, R._col, R._rowptr), shape=z[f"{tag}.dense"].shape).toarray()
    assert np.max(np.abs(dense - z[f"{tag}.dense"])) <= 1e-7 * np.max(np.abs(z[f"{tag}.dense"]))
    x = ift.makeField(R.domain, z[f"{tag}.x"], device_id)
    y = ift.makeField(R.target, z[f"{tag}.y"], device_id)
    rx, ry = R(x), R.adjoint(y)
    assert rx.device_id == device_id and ry.device_id == device_id
    assert gl.relerr(rx.asnumpy(), z[f"{tag}.times"]) < 1e-12
    assert gl.relerr(ry.asnumpy(), z[f"{tag}.adjoint"]) < 1e-12
    # adjointness (reference extra.py:220-231) incl. fp32 fields (float32 weights, fp64 accumulation)
    x32 = ift.makeField(R.domain, z[f"{tag}.x"].astype(np.float32), device_id)
    y32 = ift.makeField(R.target, z[f"{tag}.y"].astype(np.float32), device_id)
    a, b = float(R(x32).s_vdot(y32)), float(x32.s_vdot(R.adjoint(y32)))
    assert abs(a - b) < 1e-5 * max(abs(a), abs(b), 1e-30)


@pytest.mark.parametrize("tag", CASES)
def test_los_response_host(tag):
    _check_los(tag, -1)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_los_response_device(tag):
    _check_los(tag, 0)


def _check_mask(device_id):
    z = gl.load("los")
    sp = ift.RGSpace(z["mask.flags"].shape)
    M = ift.MaskOperator(ift.makeField(sp, z["mask.flags"]))
    assert M.target.shape == (int(np.logical_not(z["mask.flags"]).sum()),)
    assert np.array_equal(M(ift.makeField(M.domain, z["mask.x"], device_id)).asnumpy(), z["mask.times"])
    assert np.array_equal(M.adjoint(ift.makeField(M.target, z["mask.y"], device_id)).asnumpy(), z["mask.adjoint"])


def test_mask_operator_host():
    _check_mask(-1)


@pytest.mark.gpu
def test_mask_operator_device():
    _check_mask(0)


def test_los_argument_errors():
    sp = ift.RGSpace((8, 8))
    s, e = np.zeros((2, 3)), np.ones((2, 3))
    with pytest.raises(TypeError):
        ift.LOSResponse(ift.UnstructuredDomain(5), s, e)
    with pytest.raises(TypeError):
        ift.LOSResponse(sp, np.zeros((3, 3)), np.ones((3, 3)))
    with pytest.raises(TypeError):
        ift.LOSResponse(sp, s, np.ones((2, 4)))
    with pytest.raises(ValueError):  # 1/len - truncation*sigma < 0 (los_response.py:169-171)
        ift.LOSResponse(sp, s, e, sigmas=np.full(3, 10.0))
    with pytest.raises(TypeError):
        ift.MaskOperator(np.zeros(4))


def _c4_model(z, device_id):
    shape = (16, 16)
    sp = ift.RGSpace(shape)
    cfm = ift.CorrelatedFieldMaker("")
    cfm.add_fluctuations(sp, CF_ARGS["fluctuations"], CF_ARGS["flexibility"], CF_ARGS["asperity"], CF_ARGS["loglogavgslope"])
    cfm.set_amplitude_total_offset(2.0, (1e-1, 3e-2))
    cf = cfm.finalize()
    R = ift.LOSResponse(sp, z["c4.starts"], z["c4.ends"])
    Mk = ift.MaskOperator(ift.makeField(R.target, z["c4.flags"]))
    resp = Mk @ R @ cf.ptw("sigmoid")
    d = ift.makeField(resp.target, z["c4.data"], device_id)
    lh = ift.GaussianEnergy(d, ift.ScalingOperator(resp.target, 1.0 / 1e-3, np.float64)) @ resp
    return cf, resp, lh


def _check_c4(device_id, fuse=True):
    z = gl.load("los")
    cf, resp, lh = _c4_model(z, device_id)
    x = ift.MultiField.from_raw(cf.domain, gl.latent(z, "c4.x"), device_id)
    v = ift.MultiField.from_raw(cf.domain, gl.latent(z, "c4.v"), device_id)
    assert gl.relerr(resp(x).asnumpy(), z["c4.resp"]) < 1e-11
    ic = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=6)
    ham = ift.StandardHamiltonian(lh, ic, prior_sampling_dtype=np.float64)
    hl = ham(ift.Linearization.make_var(x, want_metric=True))
    assert abs(float(hl.val.asnumpy()) - float(z["c4.ham_value"])) < 1e-10 * abs(float(z["c4.ham_value"]))
    assert gl.lat_relerr(hl.gradient.asnumpy(), gl.latent(z, "c4.ham_grad")) < 1e-9
    assert gl.lat_relerr(hl.metric(v).asnumpy(), gl.latent(z, "c4.ham_metric_v")) < 1e-9
    ift.random.push_sseq_from_seed(44)
    try:
        mk = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=2),  # noqa: E731
                                    max_cg_iterations=8)
        nl = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, iteration_limit=3, convergence_level=2),  # noqa: E731
                                    max_cg_iterations=6)
        nit = 2 if device_id < 0 else 1
        sl, mean = ift.optimize_kl(lh, nit, 1, mk, ic, nonlinear_sampling_minimizer=nl, output_directory=None,
                                   return_final_position=True, initial_position=x, device_id=device_id, fuse=fuse)
    finally:
        ift.random.pop_sseq()
    if device_id < 0:
        # geoVI: bounded CG lengths keep the run in the reproducible regime (see make_golden.py GEO_CG)
        assert gl.lat_relerr(mean.asnumpy(), gl.latent(z, "c4.okl_mean")) < 2e-3
        for i, s in enumerate(sl.iterator()):
            assert gl.lat_relerr(s.asnumpy(), gl.latent(z, f"c4.okl_sample{i}")) < 5e-3
        return
    # device: ONE iteration (okl1.npz) -- rounding differences between the device and the host arithmetic (1e-16) are
    # amplified to 1e-3..1e-2 by a discrete decision in the second iteration of this configuration (see
    # make_golden.py::okl1_cases); since round 3 every device sum of this model is built in a fixed order (nk_csr_rowsum)
    z1 = gl.load("okl1")
    lat = lambda pre: {k[len(pre) + 1:]: np.asarray(z1[k]) for k in z1.files if k.startswith(pre + ".")}  # noqa: E731
    assert gl.lat_relerr(mean.asnumpy(), lat("c4.mean")) < 1e-6
    for i, s in enumerate(sl.iterator()):
        assert gl.lat_relerr(s.asnumpy(), lat(f"c4.sample{i}")) < 1e-6


def test_config4_model_host():
    _check_c4(-1)


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", [True, False])
def test_config4_model_device(fuse):
    """fuse=True: the fusion pass of optimize_kl recognises Mask @ LOSResponse @ sigmoid(cf) and runs the iteration on the
    fused response engine; fuse=False: the generic operator graph.  Both against the reference's golden run."""
    _check_c4(0, fuse)


@pytest.mark.gpu
def test_spmv_kernels_against_scipy_on_a_larger_response():
    """nk_spmv / nk_spmv_t on a 2000-line response over a 256 x 192 grid (rows of a few hundred entries, many lines per
    pixel) against scipy.sparse on the host; fp64 and fp32 fields."""
    from scipy.sparse import csr_matrix

    rng = np.random.default_rng(8)
    sp = ift.RGSpace((256, 192), (1.0 / 256, 1.0 / 192))
    R = ift.LOSResponse(sp, rng.uniform(size=(2, 2000)), rng.uniform(size=(2, 2000)))
    m = csr_matrix((R._wgt, R._col, R._rowptr), shape=(2000, sp.size))
    x, y = rng.normal(size=sp.shape), rng.normal(size=2000)
    for dt, tol in ((np.float64, 1e-12), (np.float32, 2e-6)):
        got = R(ift.makeField(R.domain, x.astype(dt), 0)).asnumpy()
        ref = m @ x.astype(dt).astype(np.float64).reshape(-1)
        assert gl.relerr(got, ref) < tol
        got = R.adjoint(ift.makeField(R.target, y.astype(dt), 0)).asnumpy()
        ref = (m.T @ y.astype(dt).astype(np.float64)).reshape(sp.shape)
        assert gl.relerr(got, ref) < tol
