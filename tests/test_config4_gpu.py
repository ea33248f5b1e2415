"""BASELINE config 4 on the GPU at its real size: 4096^2 RGSpace, signal = sigmoid(correlated field), data =
MaskOperator(LOSResponse(10^4 random lines)) + Gaussian noise 1e-3, geoVI with 4 samples -- the recipe of
reference demos/cl/getting_started_3.py:48-51,98-100 (library/los_response.py:144-253, operators/mask_operator.py:37-58).

 (i)   the sparse products of THAT matrix against scipy.sparse on the host (fp64, 1e-12), both directions, bit-reproducible;
 (ii)  full-size properties of the fused response engine: self-adjoint metric, linearity, gradient against a central
       finite difference, and agreement with the generic operator graph on the same device;
 (iii) the same lines on a 512^2 grid against the numpy oracle (value / gradient / metric <= 1e-9, a geoVI sample with
       bounded CG lengths <= 1e-6);
 (iv)  one geoVI optimize_kl iteration with 4 samples at full size on the fused engine and on the generic graph.
"""
import numpy as np
import pytest
import torch

import nifty_amd as ift
from oracle import nifty_oracle as orc
from tests import goldenlib as gl

pytestmark = pytest.mark.gpu

N_LOS = 10000
CF = dict(offset_mean=2.0, offset_std=(1e-1, 3e-2), fluctuations=(1.0, 5e-1), loglogavgslope=(-3.0, 2e-1),
          flexibility=(1.0, 2e-1), asperity=(5e-1, 5e-2))
NOISE_VAR = 1e-3


def _lines():
    rng = np.random.default_rng(1)
    starts, ends = rng.uniform(size=(2, N_LOS)), rng.uniform(size=(2, N_LOS))
    flags = np.zeros(N_LOS, dtype=bool)
    flags[rng.integers(0, N_LOS, N_LOS // 20)] = True
    return starts, ends, flags


def _api_model(n, device_id=0, seed=42):
    """The config through the nifty.cl-shaped API: (cf, response operator, likelihood, data, LOSResponse, mask)."""
    starts, ends, flags = _lines()
    sp = ift.RGSpace((n, n))
    cfm = ift.CorrelatedFieldMaker("")
    cfm.add_fluctuations(sp, CF["fluctuations"], CF["flexibility"], CF["asperity"], CF["loglogavgslope"])
    cfm.set_amplitude_total_offset(CF["offset_mean"], CF["offset_std"])
    cf = cfm.finalize()
    R = ift.LOSResponse(sp, starts, ends)
    Mk = ift.MaskOperator(ift.makeField(R.target, flags))
    resp = Mk @ R @ cf.ptw("sigmoid")
    ift.random.push_sseq_from_seed(seed)
    try:
        truth = ift.from_random(cf.domain, device_id=device_id)
        d = resp(truth) + ift.from_random(resp.target, device_id=device_id) * np.sqrt(NOISE_VAR)
    finally:
        ift.random.pop_sseq()
    lh = ift.GaussianEnergy(d, ift.ScalingOperator(resp.target, 1.0 / NOISE_VAR, np.float64)) @ resp
    return cf, resp, lh, d, R, Mk


@pytest.fixture(scope="module")
def full():
    return _api_model(4096)


def _fused(lh, dtype=torch.float64):
    from nifty_amd.engine import FusedModel
    from nifty_amd.optimize_kl import match_fused

    kw = match_fused(lh)
    assert kw is not None and kw["response"] is not None, "config 4 must be recognised by the fusion pass"
    return FusedModel(kw.pop("shape"), kw.pop("distances"), dtype=dtype, device="cuda:0", **kw)


def test_sparse_products_of_the_full_matrix_against_scipy(full):
    from scipy.sparse import csr_matrix

    cf, resp, lh, d, R, Mk = full
    n_pix = R.domain.size
    m = csr_matrix((R._wgt, R._col, R._rowptr), shape=(N_LOS, n_pix))
    assert m.nnz > 2e7  # ~2.7e7 entries: 10^4 lines x O(4096) pixels
    rng = np.random.default_rng(5)
    x, y = rng.normal(size=R.domain.shape), rng.normal(size=N_LOS)
    for dt, tol in ((np.float64, 1e-12), (np.float32, 2e-6)):
        xf, yf = ift.makeField(R.domain, x.astype(dt), 0), ift.makeField(R.target, y.astype(dt), 0)
        got, got_t = R(xf), R.adjoint(yf)
        assert gl.relerr(got.asnumpy(), m @ x.astype(dt).astype(np.float64).reshape(-1)) < tol
        assert gl.relerr(got_t.asnumpy().reshape(-1), m.T @ y.astype(dt).astype(np.float64)) < tol
        # fixed summation order: the same bits on every call, in both directions
        assert torch.equal(R(xf).val, got.val) and torch.equal(R.adjoint(yf).val, got_t.val)
    # the masked matrix of the fused engine = the kept rows
    from nifty_amd.los_response import SparseResponse

    sr = SparseResponse.from_operators(R, Mk)
    keep = Mk._keep.numpy()
    xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y[keep]).cuda()
    assert gl.relerr(sr.times(xt).cpu().numpy(), (m @ x.reshape(-1))[keep]) < 1e-12
    assert gl.relerr(sr.adjoint(yt).cpu().numpy(), m[keep].T @ y[keep]) < 1e-12


def test_full_size_engine_properties_and_generic_graph_agreement(full):
    from nifty_amd import random
    from nifty_amd.engine import LatentVec

    cf, resp, lh, d, R, Mk = full
    model = _fused(lh)
    random.push_sseq_from_seed(7)
    try:
        x = model.draw_prior() * 0.1
        u, v = model.draw_prior(), model.draw_prior()
    finally:
        random.pop_sseq()
    lp = model.linearize(x)
    mu, mv = model.metric(lp, u), model.metric(lp, v)
    # self-adjoint: <v, M u> = <u, M v>
    a, b = v.s_vdot(mu), u.s_vdot(mv)
    assert abs(a - b) < 1e-10 * max(abs(a), abs(b))
    # positive: <u, M u> >= <u, u> (prior metric = 1)
    assert u.s_vdot(mu) >= u.s_vdot(u) * (1 - 1e-12)
    # linear: M(2u - 3v) = 2 Mu - 3 Mv
    comb = model.metric(lp, u * 2.0 - v * 3.0)
    ref = mu * 2.0 - mv * 3.0
    diff = comb - ref
    assert diff.norm() < 1e-10 * ref.norm()
    # gradient against a central finite difference of the value along u
    eps = 1e-5
    vp = float(model.linearize(x.axpy(eps, u)).value.item())
    vm = float(model.linearize(x.axpy(-eps, u)).value.item())
    fd, an = (vp - vm) / (2 * eps), lp.grad.s_vdot(u)
    assert abs(fd - an) < 2e-6 * max(abs(an), 1.0), (fd, an)
    # the same numbers from the generic operator graph on the device (every operation in libniftyk)
    from nifty_amd.optimize_kl import _latent_to_mf

    xm, um = _latent_to_mf(lh.domain, x, np.float64), _latent_to_mf(lh.domain, u, np.float64)
    ham = ift.StandardHamiltonian(lh, ift.AbsDeltaEnergyController(0.05, iteration_limit=5),
                                  prior_sampling_dtype=np.float64)
    hl = ham(ift.Linearization.make_var(xm, want_metric=True))
    val_g = float(hl.val.asnumpy())
    assert abs(float(lp.value.item()) - val_g) < 1e-10 * abs(val_g)
    assert gl.lat_relerr(lp.grad.to_dict(), hl.gradient.asnumpy()) < 1e-9
    assert gl.lat_relerr(mu.to_dict(), hl.metric(um).asnumpy()) < 1e-9
    del LatentVec


def test_same_lines_on_512_squared_against_the_oracle():
    from nifty_amd import random
    from nifty_amd.engine import LatentVec, draw_samples
    from nifty_amd.minimization import AbsDeltaEnergyController, NewtonCG

    n = 512
    starts, ends, flags = _lines()
    cf, resp, lh, d, R, Mk = _api_model(n)
    model = _fused(lh)
    # oracle: its own walk of the lines (pure Python), float32 weights, masked rows removed
    mat = orc.los_sparse((n, n), (1.0 / n, 1.0 / n), starts, ends)[np.logical_not(flags)]
    from scipy.sparse import csr_matrix

    mine = csr_matrix((R._wgt, R._col, R._rowptr), shape=(N_LOS, n * n))[np.logical_not(flags)]
    assert abs(mine - mat).max() <= 1e-7 * abs(mat).max()
    ocf = orc.CFModel((n, n), None, orc.CFParams(**CF))
    olh = orc.Likelihood("gaussian", d.asnumpy(), icov=1.0 / NOISE_VAR, nonlin="sigmoid", response=mat)
    rng = np.random.default_rng(3)
    x = {k: 0.1 * a for k, a in ocf.draw_latent(rng).items()}
    v = ocf.draw_latent(rng)
    lin = orc.Linearized(ocf, olh, x)
    val, grad = lin.value_grad()
    xl, vl = LatentVec.from_dict(model, x), LatentVec.from_dict(model, v)
    lp = model.linearize(xl)
    assert abs(float(lp.value.item()) - val) < 1e-10 * abs(val)
    assert gl.lat_relerr(lp.grad.to_dict(), grad) < 1e-9
    assert gl.lat_relerr(model.metric(lp, vl).to_dict(), lin.metric(v)) < 1e-9
    # one geoVI sample pair, bounded CG lengths (long ill-conditioned runs amplify rounding chaotically: DESIGN 6)
    ic = lambda: AbsDeltaEnergyController(0.05, iteration_limit=6)  # noqa: E731
    oic = lambda: orc.AbsDeltaEnergyController(0.05, iteration_limit=6)  # noqa: E731
    random.push_sseq_from_seed(11)
    try:
        geo = NewtonCG(AbsDeltaEnergyController(0.5, iteration_limit=2, convergence_level=2), max_cg_iterations=6)
        res, negs, n_total = draw_samples(model, xl, 1, True, ic, geo_minimizer=geo)
    finally:
        random.pop_sseq()
    ogeo = lambda e: orc.newton_cg(e, orc.AbsDeltaEnergyController(0.5, iteration_limit=2, convergence_level=2),  # noqa: E731
                                   max_cg_iterations=6)
    ores, onegs = orc.draw_samples(ocf, olh, x, 1, True, np.random.SeedSequence(11), oic, geo_minimizer=ogeo)
    assert n_total == 2 and len(res) == len(ores) == 2
    for r, o in zip(res, ores):
        assert gl.lat_relerr(r.to_dict(), o) < 1e-6


def test_fp32_fields_behind_the_response_against_the_fp64_oracle(monkeypatch):
    """The 512^2 model with FP32 fields: the forward transform, the response, the residual and the energy run in fp64 with the
    fp32 arrays at both ends (FusedModel.wide_response) -- what the reference's promoted arithmetic does -- so value and
    gradient stay inside the 1e-5 of north_star; A/B against the all-fp32 evaluation (NK_WIDE_FORWARD=0) on the same inputs."""
    from nifty_amd.engine import LatentVec

    n = 512
    starts, ends, flags = _lines()
    cf, resp, lh, d, R, Mk = _api_model(n)
    mat = orc.los_sparse((n, n), (1.0 / n, 1.0 / n), starts, ends)[np.logical_not(flags)]
    ocf = orc.CFModel((n, n), None, orc.CFParams(**CF))
    data32 = d.asnumpy().astype(np.float32).astype(np.float64)
    olh = orc.Likelihood("gaussian", data32, icov=1.0 / NOISE_VAR, nonlin="sigmoid", response=mat)
    rng = np.random.default_rng(4)
    x = {k: 0.1 * a for k, a in ocf.draw_latent(rng).items()}
    v = ocf.draw_latent(rng)
    x["xi"] = x["xi"].astype(np.float32).astype(np.float64)
    v["xi"] = v["xi"].astype(np.float32).astype(np.float64)
    lin = orc.Linearized(ocf, olh, x)
    val, grad = lin.value_grad()
    mv = lin.metric(v)
    errs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("NK_WIDE_FORWARD", flag)
        model = _fused(lh, torch.float32)
        assert model.wide_response == (flag == "1") and not model.wide
        lp = model.linearize(LatentVec.from_dict(model, x))
        errs[flag] = (abs(float(lp.value.item()) - val) / abs(val), gl.lat_relerr(lp.grad.to_dict(), grad),
                      gl.lat_relerr(model.metric(lp, LatentVec.from_dict(model, v)).to_dict(), mv))
        del model
    print("512^2 + LOS response, fp32 fields vs fp64 oracle (value, gradient, metric): all-fp32 %.1e %.1e %.1e | wide %.1e %.1e %.1e"
          % (errs["0"] + errs["1"]))
    assert errs["1"][0] < 1e-9 and errs["1"][1] < 1e-5 and errs["1"][2] < 2e-4
    assert errs["1"][0] < errs["0"][0]


def _run_okl(lh, fuse, ic, mk, nl):
    ift.random.push_sseq_from_seed(42)
    try:
        return ift.optimize_kl(lh, 1, 4, mk, ic, nonlinear_sampling_minimizer=nl, output_directory=None,
                               return_final_position=True, device_id=0, fuse=fuse)
    finally:
        ift.random.pop_sseq()


def test_one_geovi_iteration_at_full_size(full):
    cf, resp, lh, d, R, Mk = full
    # (a) bounded CG / Newton lengths: the fused response engine and the generic operator graph walk the same algorithm
    # with the same seeds and must agree to rounding (long ill-conditioned runs amplify 1e-16 differences chaotically
    # through their discrete decisions, DESIGN 6 -- the full recipe below differs at O(1) between ANY two arithmetics)
    ic = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=4)
    mk = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=2),  # noqa: E731
                                max_cg_iterations=4)
    nl = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, iteration_limit=2, convergence_level=2),  # noqa: E731
                                max_cg_iterations=4)
    (sl_f, mean_f), (sl_g, mean_g) = _run_okl(lh, True, ic, mk, nl), _run_okl(lh, False, ic, mk, nl)
    assert sl_f.n_samples == sl_g.n_samples == 8  # geoVI keeps both members of a mirrored pair as independent residuals
    err = gl.lat_relerr(mean_f.asnumpy(), mean_g.asnumpy())
    print(f"4096^2 geoVI iteration, bounded recipe: fused vs generic graph mean {err:.2e}")
    # (measured 4.6e-6: two Newton steps of four CG iterations each amplify the 1e-13 differences of the two arithmetics)
    assert err < 1e-4
    for a, b in zip(sl_f.iterator(), sl_g.iterator()):
        assert gl.lat_relerr(a.asnumpy(), b.asnumpy()) < 1e-4
    # (b) the recipe of the config (demos/cl/getting_started_3.py:119-127) on the fused engine: finite, and the posterior
    # mean explains the data better than the start
    ic = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=20)
    mk = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=3),  # noqa: E731
                                max_cg_iterations=20)
    nl = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, iteration_limit=5, convergence_level=2))  # noqa: E731
    sl, mean = _run_okl(lh, True, ic, mk, nl)
    assert sl.n_samples == 8
    for k in mean.keys():
        assert bool(torch.isfinite(mean[k].val).all())
    r = (resp(mean) - d).asnumpy()
    r0 = (resp(ift.full(lh.domain, 0.0).at(0)) - d).asnumpy()
    chi2, chi2_0 = float(np.mean(r * r) / NOISE_VAR), float(np.mean(r0 * r0) / NOISE_VAR)
    print(f"4096^2 geoVI iteration, full recipe: reduced chi^2 {chi2_0:.1f} -> {chi2:.1f}")
    assert chi2 < chi2_0
