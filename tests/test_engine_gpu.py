"""GPU parity tests of the fused engine (HIP path through the C ABI) against the oracle and the golden
vectors generated from the reference.  Tolerance: 1e-5 relative in fp64 is the north-star bar; the
checks below are far tighter wherever the arithmetic allows (documented per assertion)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import nifty_oracle as orc
from tests import goldenlib as gl

pytestmark = pytest.mark.gpu

NON_GEO = ("g1d", "p2d", "g3d")


def _model(z, dtype=torch.float64):
    from nifty_amd.engine import FusedModel

    m = gl.meta(z)
    kw = dict(offset_mean=2.0, likelihood=m["kind"], nonlin=m["nonlin"], dtype=dtype, device="cuda:0")
    if m["kind"] == "gaussian":
        ic = z["icov"]
        kw["icov"] = float(ic) if ic.shape == () else ic
    model = FusedModel(m["shape"], m["distances"], data=z["data"], **kw)
    return m, model


def _lv(model, d):
    from nifty_amd.engine import LatentVec

    return LatentVec.from_dict(model, d)


@pytest.mark.parametrize("case", gl.MODEL_CASES)
def test_signal_and_hamiltonian_vs_reference_golden(case):
    z = gl.load("model_" + case)
    m, model = _model(z)
    x, v = gl.latent(z, "x"), gl.latent(z, "v")
    xl, vl = _lv(model, x), _lv(model, v)
    sig = model.signal(xl).cpu().numpy()
    g, _ = orc.NONLIN[m["nonlin"]]
    assert gl.relerr(sig, g(z["cf"])) < 1e-11
    lp = model.linearize(xl)
    hv = float(lp.value.item())
    assert abs(hv - float(z["ham_value"])) < 1e-11 * abs(float(z["ham_value"]))
    assert gl.lat_relerr(lp.grad.to_dict(), gl.latent(z, "ham_grad")) < 1e-10
    mv = model.metric(lp, vl).to_dict()
    assert gl.lat_relerr(mv, gl.latent(z, "ham_metric_v")) < 1e-10


@pytest.mark.parametrize("case", NON_GEO + ("g2d_dist",))
def test_mgvi_samples_kl_newton_vs_reference_golden(case):
    from nifty_amd import random
    from nifty_amd.engine import FusedKL, draw_samples
    from nifty_amd.minimization import AbsDeltaEnergyController, NewtonCG

    z = gl.load("model_" + case)
    m, model = _model(z)
    x, v = gl.latent(z, "x"), gl.latent(z, "v")
    xl, vl = _lv(model, x), _lv(model, v)
    random.push_sseq_from_seed(m["seed"] + 1)
    try:
        res, negs, n_total = draw_samples(model, xl, m["n_samples"], True,
                                          lambda: AbsDeltaEnergyController(0.05, iteration_limit=m["sampling_limit"]))
    finally:
        random.pop_sseq()
    assert n_total == int(z["n_residuals"])
    for i, (r, neg) in enumerate(zip(res, negs)):
        rd = r.to_dict()
        if neg:
            rd = {k: -a for k, a in rd.items()}
        assert gl.lat_relerr(rd, gl.latent(z, f"residual{i}")) < 1e-8, i
    kl = FusedKL(model, xl, res, negs)
    assert abs(kl.value - float(z["kl_value"])) < 1e-9 * abs(float(z["kl_value"]))
    assert gl.lat_relerr(kl.gradient.to_dict(), gl.latent(z, "kl_grad")) < 1e-8
    assert gl.lat_relerr(kl.apply_metric(vl).to_dict(), gl.latent(z, "kl_metric_v")) < 1e-8
    mini = NewtonCG(AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=2), max_cg_iterations=8)
    kl2, _ = mini(kl)
    assert abs(kl2.value - float(z["kl_min_value"])) < 1e-7 * abs(float(z["kl_min_value"]))
    assert gl.lat_relerr(kl2.position.to_dict(), gl.latent(z, "kl_min_pos")) < 1e-6


@pytest.mark.parametrize("case", ["p2d_geo", "g2d_sig_geo"])
def test_geovi_samples_vs_reference_golden(case):
    """geoVI inside the fused engine (FusedGeoEnergy, kl_energies.py:105-124, 148-155) against the reference's samples,
    KL value / gradient / metric and a NewtonCG step.  Tolerances as in the generic-graph geoVI tests: the golden CG
    lengths are bounded to stay in the reproducible regime (make_golden.py GEO_CG)."""
    from nifty_amd import random
    from nifty_amd.engine import FusedKL, draw_samples
    from nifty_amd.minimization import AbsDeltaEnergyController, NewtonCG

    z = gl.load("model_" + case)
    m, model = _model(z)
    xl, vl = _lv(model, gl.latent(z, "x")), _lv(model, gl.latent(z, "v"))
    geo = NewtonCG(AbsDeltaEnergyController(0.5, iteration_limit=3, convergence_level=2), max_cg_iterations=6)
    random.push_sseq_from_seed(m["seed"] + 1)
    try:
        res, negs, n_total = draw_samples(model, xl, m["n_samples"], True,
                                          lambda: AbsDeltaEnergyController(0.05, iteration_limit=m["sampling_limit"]),
                                          geo_minimizer=geo)
    finally:
        random.pop_sseq()
    assert n_total == int(z["n_residuals"]) and not any(negs)
    for i, r in enumerate(res):
        assert gl.lat_relerr(r.to_dict(), gl.latent(z, f"residual{i}")) < 1e-6, i
    kl = FusedKL(model, xl, res, negs)
    assert abs(kl.value - float(z["kl_value"])) < 1e-6 * abs(float(z["kl_value"]))
    assert gl.lat_relerr(kl.gradient.to_dict(), gl.latent(z, "kl_grad")) < 1e-5
    assert gl.lat_relerr(kl.apply_metric(vl).to_dict(), gl.latent(z, "kl_metric_v")) < 1e-5


@pytest.mark.parametrize("shape", [(64, 128), (256,), (64, 64, 64), (64, 64, 2048), (2048, 64, 64), (30, 50), (12, 10, 14)])
def test_geovi_energy_vs_oracle(shape):
    """FusedGeoEnergy value / gradient / metric against the oracle's restatement on seeded inputs (multiply prologue,
    data-space JVP / VJP with weight fields) on the 1-D, strided-first (incl. the smallest final tiles) and generic
    mixed-radix pipelines."""
    from nifty_amd.engine import FusedGeoEnergy, FusedModel, LatentVec

    rng = np.random.default_rng(4)
    cf = orc.CFModel(shape, None, orc.CFParams(offset_mean=1.5))
    data = rng.poisson(np.exp(cf.forward(cf.draw_latent(rng)))).astype(np.int64)
    lh = orc.Likelihood("poisson", data, nonlin="exp")
    model = FusedModel(shape, offset_mean=1.5, likelihood="poisson", data=data, nonlin="exp")
    p = {k: 0.2 * a for k, a in cf.draw_latent(rng).items()}
    x = {k: 0.2 * a for k, a in cf.draw_latent(rng).items()}
    mt = cf.draw_latent(rng)
    d = cf.draw_latent(rng)
    lin_p = orc.Linearized(cf, lh, p)
    en_o = orc._GeoEnergy(cf, lh, lin_p, lh.trafo_weight(lin_p.s), mt, x)
    pl, xl, ml, dl = (LatentVec.from_dict(model, a) for a in (p, x, mt, d))
    en = FusedGeoEnergy(model, model.trafo_point(pl), ml, xl)
    assert abs(en.value - en_o.value) < 1e-10 * abs(en_o.value)
    assert gl.lat_relerr(en.gradient.to_dict(), en_o.gradient) < 1e-9
    assert gl.lat_relerr(en.apply_metric(dl).to_dict(), en_o.apply_metric(d)) < 1e-9


@pytest.mark.parametrize("shape,kind,nonlin", [((256,), "gaussian", None), ((128, 64), "poisson", "exp"),
                                               ((32, 16, 64), "gaussian", "sigmoid"), ((64, 64, 64), "gaussian", None),
                                               # non-power-of-two grids (mixed radix 2/3/5/7, generic kernels)
                                               ((30, 50), "gaussian", None), ((12, 10, 14), "poisson", "exp"),
                                               ((768, 768), "gaussian", None), ((96, 160, 96), "gaussian", "sigmoid"),
                                               # strided-first pipeline with unequal axes
                                               ((128, 64, 256), "gaussian", None),
                                               # long last axis: the final pass runs its smallest tiles (one line pair
                                               # per workgroup; the VJP keeps the couple (b0, M - b0) together)
                                               ((64, 64, 1024), "gaussian", None), ((64, 64, 2048), "poisson", "exp"),
                                               ((64, 64, 4096), "gaussian", "sigmoid"),
                                               # long strided axes (largest register-resident line lengths)
                                               ((2048, 64, 64), "gaussian", None), ((64, 4096, 64), "poisson", "exp"),
                                               ((4096, 64, 128), "gaussian", None), ((4096, 1024), "gaussian", "sigmoid"),
                                               ((512, 4096), "poisson", "exp")])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_engine_vs_oracle_seeded(shape, kind, nonlin, dtype):
    """Same seeded inputs through the HIP engine and the numpy oracle; fp32 fields use fp64 accumulators."""
    from nifty_amd.engine import FusedModel, LatentVec

    rng = np.random.default_rng(11)
    cf = orc.CFModel(shape, None, orc.CFParams(offset_mean=1.5))
    truth = cf.draw_latent(rng)
    s = cf.forward(truth)
    g, _ = orc.NONLIN[nonlin]
    if kind == "gaussian":
        data = g(s) + 0.1 * rng.normal(size=shape)
        lh = orc.Likelihood("gaussian", data, icov=100.0, nonlin=nonlin)
        model = FusedModel(shape, offset_mean=1.5, likelihood="gaussian", data=data, icov=100.0, nonlin=nonlin, dtype=dtype)
    else:
        data = rng.poisson(g(s)).astype(np.int64)
        lh = orc.Likelihood("poisson", data, nonlin=nonlin)
        model = FusedModel(shape, offset_mean=1.5, likelihood="poisson", data=data, nonlin=nonlin, dtype=dtype)
    x = {k: 0.3 * a for k, a in cf.draw_latent(rng).items()}
    v = cf.draw_latent(rng)
    if dtype == torch.float32:  # identical inputs for both paths
        x["xi"] = x["xi"].astype(np.float32).astype(np.float64)
        v["xi"] = v["xi"].astype(np.float32).astype(np.float64)
        if kind == "gaussian":
            lh.data = lh.data.astype(np.float32).astype(np.float64)
    lin = orc.Linearized(cf, lh, x)
    val, grad = lin.value_grad()
    mv = lin.metric(v)
    xl, vl = LatentVec.from_dict(model, x), LatentVec.from_dict(model, v)
    lp = model.linearize(xl)
    # fp32 fields: value + gradient within the 1e-5 of north_star on every grid (their forward transform runs in fp64:
    # FusedModel.wide on the register-resident pipelines, FusedModel.wide_generic on mixed-radix grids and short axes);
    # a metric application is two fp32 transforms with a pointwise weight in between
    if dtype == torch.float32:
        assert model.wide or model.wide_generic
    # (12 x 10 x 14 Poisson: 1.8e-5 -- the residual lambda - d cancels, and on 1680 points the fp32 ADJOINT transform's rounding,
    # relative to lambda, is not averaged down; the generic Gaussian cases and every register-resident case hold 1e-5)
    # metric application, fp32: asserted at twice the worst error measured per class on the GPU (round 6,
    # gpurun_out/r06h/seeded_errors.txt -> profiles/r06_seeded_errors.txt): 1.4e-6 over all cases but one -- 96 x 160 x 96 with
    # the sigmoid, 5.6e-5 PER KEY: the key is the scalar `fluctuations`, a cancelling sum over all bins that comes out at 30
    # where `xi` reaches 2.3e4 (7e-8 of the vector's largest entry; every other key of that case <= 5e-7) -- until round 5 a
    # blanket 2e-4 for every fp32 case
    mtol32 = 1.2e-4 if (shape == (96, 160, 96) and nonlin == "sigmoid") else 3e-6
    tol, mtol = (1e-10, 1e-10) if dtype == torch.float64 else (1e-5 if (model.wide or kind == "gaussian") else 5e-5, mtol32)
    e_val = abs(float(lp.value.item()) - val) / abs(val)
    e_grad, e_met = gl.lat_relerr(lp.grad.to_dict(), grad), gl.lat_relerr(model.metric(lp, vl).to_dict(), mv)
    print(f"{shape} {kind} {nonlin} {dtype}: value {e_val:.1e} gradient {e_grad:.1e} metric {e_met:.1e}")
    assert e_val < tol and e_grad < tol and e_met < mtol
    # adjointness of the metric (reference extra.py:220-231): <u, M v> == <M u, v>
    u = LatentVec.from_dict(model, cf.draw_latent(rng))
    mvl = model.metric(lp, vl)
    a = u.s_vdot(mvl)
    b = model.metric(lp, u).s_vdot(vl)
    # <u, M v> of two random vectors cancels to a small number: the rounding scale is |u| |M v|, not |a|
    assert abs(a - b) < (1e-12 if dtype == torch.float64 else 2e-6) * u.norm() * mvl.norm()


def test_engine_rejects_cpu():
    from nifty_amd.engine import FusedModel

    with pytest.raises(RuntimeError):
        FusedModel((16,), data=np.zeros(16), device="cpu")


@pytest.mark.parametrize("shape,kind,nonlin,dtype", [
    ((1024, 1024, 1024), "gaussian", None, torch.float32),   # BASELINE configs[4] (headline), one sample's worth
    ((512, 512, 512), "gaussian", None, torch.float64),      # configs[2]
    ((2048, 2048), "poisson", "exp", torch.float64),         # configs[1]
    ((4096, 4096), "gaussian", "sigmoid", torch.float64)])   # configs[3] grid
def test_engine_full_size_properties(shape, kind, nonlin, dtype):
    """BASELINE.json's full sizes, through properties that need no CPU reference (reference extra.py:220-231 adjointness /
    linearity checks, extra.py:354-380 Jacobian consistency): the metric is linear, self-adjoint and positive, and the
    gradient is the derivative of the value along a random direction."""
    from nifty_amd.engine import FusedModel

    dev = torch.device("cuda:0")
    model = FusedModel(shape, offset_mean=1.5, likelihood=kind, icov=100.0, nonlin=nonlin, dtype=dtype, device=dev)
    gen = torch.Generator(device=dev).manual_seed(5)
    truth = model.draw_prior(gen) * 0.3
    s = model.signal(truth)
    if kind == "poisson":
        model.set_data(torch.poisson(s.to(torch.float64), generator=gen).to(torch.int64))
    else:
        model.set_data(s + 0.1 * torch.randn(shape, dtype=dtype, device=dev, generator=gen), 100.0)
    del s
    x = model.draw_prior(gen) * 0.2
    u, v = model.draw_prior(gen), model.draw_prior(gen)
    lp = model.linearize(x)
    f32 = dtype == torch.float32
    # self-adjoint, positive
    mv = model.metric(lp, v)
    a, b = u.s_vdot(mv), model.metric(lp, u).s_vdot(v)
    assert abs(a - b) < (2e-4 if f32 else 1e-10) * max(abs(a), abs(b), abs(v.s_vdot(mv)))
    assert v.s_vdot(mv) > 0.0
    # linear: M(2u - 3v) = 2 Mu - 3 Mv
    lhs = model.metric(lp, u * 2.0 - v * 3.0)
    rhs = model.metric(lp, u) * 2.0 - mv * 3.0
    assert (lhs - rhs).norm() < (2e-5 if f32 else 1e-11) * rhs.norm()
    del lhs, rhs, mv
    # gradient = derivative of the value (central difference along v; fp32 fields limit the step size)
    eps = 1e-2 if f32 else 1e-5
    hp = float(model.linearize(x + v * eps).value.item())
    hm = float(model.linearize(x - v * eps).value.item())
    slope = lp.grad.s_vdot(v)
    assert abs((hp - hm) / (2 * eps) - slope) < (5e-3 if f32 else 1e-6) * max(abs(slope), abs(hp) * 1e-6 / eps)


@pytest.mark.parametrize("shape,dtype,lh", [((64, 128), torch.float64, "gaussian"), ((64, 64, 128), torch.float64, "gaussian"),
                                            ((128, 256), torch.float64, "poisson"), ((64, 64, 128), torch.float32, "gaussian")])
def test_sandwich_engine_vs_oracle(shape, dtype, lh, monkeypatch):
    """Grids large enough for the five-pass sandwich (every axis >= 64, last >= 128): metric application, MGVI sample
    (CG with the fused curvature dot AND the direction update riding in the first transform pass) and KL against the
    numpy oracle; and the same run with the fusions switched off."""
    from nifty_amd import random
    from nifty_amd.engine import FusedKL, FusedModel, LatentVec, draw_samples
    from nifty_amd.minimization import AbsDeltaEnergyController

    rng = np.random.default_rng(3)
    cf = orc.CFModel(shape, None, orc.CFParams(offset_mean=1.0))
    x = {k: 0.2 * v for k, v in cf.draw_latent(rng).items()}
    v = cf.draw_latent(rng)
    if lh == "poisson":
        data = rng.poisson(np.exp(cf.forward(cf.draw_latent(rng)) * 0.3)).astype(np.int64)
        lho = orc.Likelihood("poisson", data, nonlin="exp")
        kw = dict(likelihood="poisson", data=data, nonlin="exp")
    else:
        data = cf.forward(cf.draw_latent(rng)) + 0.1 * rng.normal(size=shape)
        lho = orc.Likelihood("gaussian", data, icov=100.0)
        kw = dict(likelihood="gaussian", data=data, icov=100.0)
    lin = orc.Linearized(cf, lho, x)
    mv = lin.metric(v)
    tol = 1e-9 if dtype == torch.float64 else 2e-4

    def rel(a, b):
        return max(float(np.max(np.abs(a[k] - b[k])) / max(np.max(np.abs(b[k])), 1e-30)) for k in ("xi", "spectrum"))

    results = []
    for sandwich, fused_dir, recur in (("2", "1", "0"), ("2", "0", "0"), ("0", "0", "0"), ("2", "1", "1")):
        monkeypatch.setenv("NK_SANDWICH", sandwich)
        monkeypatch.setenv("NK_CG_FUSED_DIRECTION", fused_dir)
        monkeypatch.setenv("NK_CG_ENERGY_RECURRENCE", recur)  # 1: energy of the CG iterate by recurrence (nk_cg_update_dr)
        model = FusedModel(shape, offset_mean=1.0, dtype=dtype, device="cuda:0", **kw)
        assert model.sandwich == (sandwich == "2") and model.fused_direction == (fused_dir == "1")
        xl, vl = LatentVec.from_dict(model, x), LatentVec.from_dict(model, v)
        lp = model.linearize(xl)
        assert rel(model.metric(lp, vl).to_dict(), mv) < tol
        random.push_sseq_from_seed(7)
        res, negs, _ = draw_samples(model, xl, 1, True, lambda: AbsDeltaEnergyController(0.05, iteration_limit=6))
        random.pop_sseq()
        kl = FusedKL(model, xl, res, negs)
        results.append((res[0].to_dict(), kl.value, kl.apply_metric(vl).to_dict()))
    res_o, negs_o = orc.draw_samples(cf, lho, x, 1, True, np.random.SeedSequence(7),
                                     lambda: orc.AbsDeltaEnergyController(0.05, iteration_limit=6))
    kl_o = orc.SampledKL(cf, lho, x, res_o, negs_o)
    ctol = 1e-7 if dtype == torch.float64 else 5e-3  # six CG iterations amplify rounding differences
    for r0, val, mv_kl in results:
        assert rel(r0, res_o[0]) < ctol
        assert abs(val - kl_o.value) < (1e-8 if dtype == torch.float64 else 1e-4) * abs(kl_o.value)
        assert rel(mv_kl, kl_o.apply_metric(v)) < ctol
    # the fused and the separate direction update are the same arithmetic
    assert rel(results[0][0], results[1][0]) < (1e-12 if dtype == torch.float64 else 1e-5)


@pytest.mark.parametrize("shape,dtype", [((64, 64, 128), torch.float64), ((64, 128, 128), torch.float32)])
def test_pair_final_pass_is_bit_identical(shape, dtype):
    """Two samples' metric applications with their final passes in one launch (nk_hartley_sandwich_pair,
    FusedModel.lh_metric_accumulate_pair) against the two single applications: the same bits in the xi part, the spectrum
    part and the octant sums -- as the first two terms of a sum and accumulated onto an existing one; and the KL metric of
    a sample list (pairs for the middle samples) against the unpaired loop."""
    from nifty_amd import random
    from nifty_amd.engine import FusedKL, FusedModel, LatentVec

    model = FusedModel(shape, offset_mean=2.0, likelihood="gaussian", icov=100.0, dtype=dtype, device="cuda:0")
    random.push_sseq_from_seed(9)
    try:
        truth = model.draw_prior()
        model.set_data(model.signal(truth), 100.0)
        assert model.pair_ready()
        xs = [0.1 * model.draw_prior() for _ in range(5)]
        d = model.draw_prior()
    finally:
        random.pop_sseq()
    lps = [model.linearize(x) for x in xs]
    for first in (True, False):
        seq = LatentVec(torch.full_like(d.xi, 0.25), torch.full_like(d.small, 0.5))
        par = LatentVec(seq.xi.clone(), seq.small.clone())
        model.lh_metric_accumulate(lps[0], d, seq, 0.125, first)
        model.lh_metric_accumulate(lps[1], d, seq, 0.125, False)
        w8_b = model.w8.clone()
        model.lh_metric_accumulate_pair(lps[0], lps[1], d, par, 0.125, first)
        assert torch.equal(seq.xi, par.xi) and torch.equal(seq.small, par.small)
        assert torch.equal(w8_b, model._pair["w8"])
    # the KL metric over five samples: first and last single (direction update / identity + curvature), the middle ones paired
    res = [x - xs[0] for x in xs]
    kl = FusedKL(model, xs[0], res, [False] * 5)
    paired = kl.apply_metric(d)
    os.environ["NK_PAIR_FINAL"] = "0"
    try:
        single = kl.apply_metric(d)
    finally:
        del os.environ["NK_PAIR_FINAL"]
    assert torch.equal(paired.xi, single.xi) and torch.equal(paired.small, single.small)
    # a Newton-CG step on the KL (the CG rides its direction update in the first sample's first pass, the identity term and
    # the curvature dot in the last sample's epilogue): pairs (0,1), (2,3) and a single, or two pairs, against single launches
    from nifty_amd.minimization import AbsDeltaEnergyController, NewtonCG

    for n in (5, 4):
        results = []
        for flag in ("1", "0"):
            os.environ["NK_PAIR_FINAL"] = flag
            try:
                kln = FusedKL(model, xs[0], res[:n], [False] * n)
                mini = NewtonCG(AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=2), max_cg_iterations=6)
                e, _ = mini(kln)
                results.append((e.position.xi.clone(), e.position.small.clone(), e.value))
            finally:
                del os.environ["NK_PAIR_FINAL"]
        assert torch.equal(results[0][0], results[1][0]) and torch.equal(results[0][1], results[1][1])
        assert results[0][2] == results[1][2]


def test_fp32_fields_on_a_generic_grid_across_lanes(monkeypatch):
    """fp32 fields on a mixed-radix grid (48 x 80: generic kernels): every lane of a KL evaluation owns its fp64 copies for
    the wide forward transform (FusedModel.wide_generic; the Gaussian data in fp64 is shared) -- the lanes' sample sum agrees
    with the one-stream evaluation to fp32 rounding of the sum, and a second evaluation reproduces the first bit for bit."""
    from nifty_amd import random
    from nifty_amd.engine import FusedKL, FusedModel, LatentVec, draw_samples
    from nifty_amd.minimization import AbsDeltaEnergyController

    shape = (48, 80)
    rng = np.random.default_rng(8)
    cf = orc.CFModel(shape, None, orc.CFParams(offset_mean=1.0))
    x = {k: 0.2 * v for k, v in cf.draw_latent(rng).items()}
    data = cf.forward(cf.draw_latent(rng)) + 0.1 * rng.normal(size=shape)
    monkeypatch.setenv("NK_LANES", "0")
    model = FusedModel(shape, offset_mean=1.0, dtype=torch.float32, device="cuda:0", likelihood="gaussian", data=data, icov=100.0)
    assert model.wide_generic and not model.octant_vjp
    xl = LatentVec.from_dict(model, x)
    random.push_sseq_from_seed(12)
    res, negs, _ = draw_samples(model, xl, 2, True, lambda: AbsDeltaEnergyController(0.05, iteration_limit=3))
    random.pop_sseq()
    ref = FusedKL(model, xl, res, negs)
    assert len(ref._lanes) == 1
    ref_g = ref.gradient.to_dict()
    monkeypatch.setenv("NK_LANES", "4")
    kl = FusedKL(model, xl, res, negs)
    assert len(kl._lanes) == 4 and all(lane.wide_generic for lane in kl._lanes)
    assert abs(kl.value - ref.value) < 1e-6 * abs(ref.value)
    got = kl.gradient.to_dict()
    for k in ref_g:
        assert np.max(np.abs(got[k] - ref_g[k])) <= 1e-5 * max(np.max(np.abs(ref_g[k])), 1e-30)
    again = FusedKL(model, xl, res, negs)
    assert again.value == kl.value and all(np.array_equal(again.gradient.to_dict()[k], got[k]) for k in got)


@pytest.mark.parametrize("shape,lh", [((128, 256), "poisson"), ((64, 64, 128), "gaussian")])
def test_sample_lanes_agree_with_the_single_stream_path(shape, lh, monkeypatch):
    """Small grids: the chains of the local samples of a KL run side by side on several streams (FusedModel.lanes).  Same
    linearisations, the sample sum grouped per lane: value, gradient and metric application agree with the one-stream
    evaluation to rounding -- for every lane count, with more samples than lanes and an odd count."""
    from nifty_amd import random
    from nifty_amd.engine import FusedKL, FusedModel, LatentVec, draw_samples
    from nifty_amd.minimization import AbsDeltaEnergyController

    rng = np.random.default_rng(5)
    cf = orc.CFModel(shape, None, orc.CFParams(offset_mean=1.0))
    x = {k: 0.2 * v for k, v in cf.draw_latent(rng).items()}
    v = cf.draw_latent(rng)
    if lh == "poisson":
        data = rng.poisson(np.exp(cf.forward(cf.draw_latent(rng)) * 0.3)).astype(np.int64)
        kw = dict(likelihood="poisson", data=data, nonlin="exp")
    else:
        data = cf.forward(cf.draw_latent(rng)) + 0.1 * rng.normal(size=shape)
        kw = dict(likelihood="gaussian", data=data, icov=100.0)
    monkeypatch.setenv("NK_LANES", "0")
    model = FusedModel(shape, offset_mean=1.0, dtype=torch.float64, device="cuda:0", **kw)
    xl, vl = LatentVec.from_dict(model, x), LatentVec.from_dict(model, v)
    random.push_sseq_from_seed(11)
    res, negs, _ = draw_samples(model, xl, 3, True, lambda: AbsDeltaEnergyController(0.05, iteration_limit=3))
    random.pop_sseq()
    res, negs = res[:5], negs[:5]  # an odd number of samples
    ref = FusedKL(model, xl, res, negs)
    assert len(ref._lanes) == 1
    ref_m = ref.apply_metric(vl).to_dict()
    ref_g = ref.gradient.to_dict()
    for lanes in ("2", "4", "8"):
        monkeypatch.setenv("NK_LANES", lanes)
        kl = FusedKL(model, xl, res, negs)
        assert len(kl._lanes) == min(int(lanes), 5)
        assert abs(kl.value - ref.value) < 1e-13 * abs(ref.value)
        got_g, got_m = kl.gradient.to_dict(), kl.apply_metric(vl).to_dict()
        for k in ref_g:
            assert np.max(np.abs(got_g[k] - ref_g[k])) <= 1e-12 * max(np.max(np.abs(ref_g[k])), 1e-30)
            assert np.max(np.abs(got_m[k] - ref_m[k])) <= 1e-12 * max(np.max(np.abs(ref_m[k])), 1e-30)
        # twice the same: the lanes leave no state behind
        again = kl.apply_metric(vl).to_dict()
        assert all(np.array_equal(again[k], got_m[k]) for k in again)
        # through the minimiser's protocol (CG with separate curvature dot)
        assert not kl.metric.fused_dot
