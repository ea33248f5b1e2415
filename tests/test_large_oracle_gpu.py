"""Oracle-anchored numbers in the LARGE regime (the full-size tests elsewhere are property checks only):

 * 512^3 fp64 -- BASELINE config 3 at its real size -- value, gradient, metric application and one mirrored MGVI sample
   pair of the fused HIP path against the numpy oracle (scipy.fft on all host cores), <= 1e-9 relative;
 * 256^3 fp32 fields (fp64 accumulators, the arithmetic of the headline config) against the fp64 oracle: the measured
   error is asserted here and quoted in DESIGN.md 6 -- the north-star bar is 1e-5 relative for fp64; fp32 fields carry
   their own rounding.  Measured (round 3): a METRIC application agrees to 1e-7 .. 2e-7 of the largest entry at 256^3 and
   512^3 alike; value 1e-10; the GRADIENT to 4e-5 (256^3) / 7e-5 .. 2.5e-4 (512^3).  The gradient's error is not produced by
   the adjoint transform (the fp32 VJP of an identical residual agrees to 1.5e-7, tools/gpu_fp32_error_probe.py) but by the
   6e-8 relative rounding of a(k) xi(k) of the few dominant low-|k| modes in the FORWARD transform: a coherent error of the
   signal, which the adjoint weights with N a(k) against the sqrt(N) of the white residual -- it grows like sqrt(N) and is
   inherent to fp32 fields (the reference's fp32 path multiplies a[pindex] xi in fp32 as well).
"""
import os

import numpy as np
import pytest
import torch

from oracle import nifty_oracle as orc
from tests import goldenlib as gl

pytestmark = pytest.mark.gpu


def _setup(shape, seed):
    cores = os.cpu_count() or 1
    cf = orc.CFModel(shape, None, orc.CFParams(offset_mean=2.0), workers=cores)
    rng = np.random.default_rng(seed)
    x = {k: 0.1 * a for k, a in cf.draw_latent(rng).items()}
    v = cf.draw_latent(rng)
    data = cf.forward(x) + 0.1 * rng.normal(size=shape)
    lh = orc.Likelihood("gaussian", data, icov=100.0)
    return cf, lh, x, v, data


def _table(got, ref):
    """per key: (max |diff| / max |ref[key]|,  max |diff| / largest entry of the whole vector)"""
    scale = max(float(np.max(np.abs(ref[k]))) for k in ref)
    out = {}
    for k in ref:
        e = float(np.max(np.abs(np.asarray(got[k], dtype=np.float64) - ref[k])))
        out[k] = (e / max(float(np.max(np.abs(ref[k]))), 1e-300), e / scale)
    return out


def _show(tag, tab):
    print(tag + ": " + "  ".join(f"{k} {a:.1e}/{b:.1e}" for k, (a, b) in tab.items()))


def _errors(model, lin, x, v, tag=""):
    """(value error, gradient error, metric error): the latter two as max over keys of max |diff| relative to the LARGEST
    entry of the whole latent vector (the per-key table is printed)."""
    from nifty_amd.engine import LatentVec

    val, grad = lin.value_grad()
    mv = lin.metric(v)
    lp = model.linearize(LatentVec.from_dict(model, x))
    got_mv = model.metric(lp, LatentVec.from_dict(model, v)).to_dict()
    tg, tm = _table(lp.grad.to_dict(), grad), _table(got_mv, mv)
    _show(tag + " gradient (per key / global)", tg)
    _show(tag + " metric   (per key / global)", tm)
    return (abs(float(lp.value.item()) - val) / abs(val), max(b for _, b in tg.values()), max(b for _, b in tm.values()))


@pytest.mark.timeout(1500)
def test_config3_full_size_against_the_oracle():
    from nifty_amd import random
    from nifty_amd.engine import FusedModel, LatentVec, draw_samples
    from nifty_amd.minimization import AbsDeltaEnergyController

    shape = (512, 512, 512)
    cf, lh, x, v, data = _setup(shape, 21)
    model = FusedModel(shape, offset_mean=2.0, likelihood="gaussian", data=data, icov=100.0, dtype=torch.float64,
                       device="cuda:0")
    assert model.sandwich and model.scatter_fixed_point
    lin = orc.Linearized(cf, lh, x)
    e_val, e_grad, e_met = _errors(model, lin, x, v, "512^3 fp64")
    print(f"512^3 fp64 vs oracle: value {e_val:.2e} gradient {e_grad:.2e} metric {e_met:.2e}")
    assert e_val < 1e-11 and e_grad < 1e-9 and e_met < 1e-9
    # the same point with fp32 fields (how the fp32 error grows from 1.7e7 to 1.3e8 points)
    model32 = FusedModel(shape, offset_mean=2.0, likelihood="gaussian", data=data, icov=100.0, dtype=torch.float32,
                         device="cuda:0")
    f_val, f_grad, f_met = _errors(model32, lin, x, v, "512^3 fp32")
    print(f"512^3 fp32 vs fp64 oracle: value {f_val:.2e} gradient {f_grad:.2e} metric {f_met:.2e}")
    del model32
    assert f_val < 1e-8 and f_grad < 1e-3 and f_met < 1e-6
    # one mirrored MGVI sample pair, three CG iterations (bounded: long runs amplify rounding)
    random.push_sseq_from_seed(5)
    try:
        res, negs, n_total = draw_samples(model, LatentVec.from_dict(model, x), 1, True,
                                          lambda: AbsDeltaEnergyController(0.05, iteration_limit=3))
    finally:
        random.pop_sseq()
    ores, onegs = orc.draw_samples(cf, lh, x, 1, True, np.random.SeedSequence(5),
                                   lambda: orc.AbsDeltaEnergyController(0.05, iteration_limit=3))
    assert n_total == 2 and negs == onegs == [False, True]
    e_s = gl.lat_relerr(res[0].to_dict(), ores[0])
    print(f"512^3 fp64 MGVI sample vs oracle: {e_s:.2e}")
    # the device draws the very numpy stream (bit-identical outside the ziggurat tail, <= 4 ulp inside): the sample
    # differs from the oracle's by rounding only
    assert e_s < 1e-8


@pytest.mark.timeout(900)
def test_fp32_fields_against_the_fp64_oracle_at_256_cubed():
    from nifty_amd.engine import FusedModel

    shape = (256, 256, 256)
    cf, lh, x, v, data = _setup(shape, 22)
    lin = orc.Linearized(cf, lh, x)
    errs = {}
    for dt in (torch.float64, torch.float32):
        model = FusedModel(shape, offset_mean=2.0, likelihood="gaussian", data=data, icov=100.0, dtype=dt, device="cuda:0")
        errs[dt] = _errors(model, lin, x, v, "256^3 " + ("fp64" if dt == torch.float64 else "fp32"))
        del model
    print("256^3 vs fp64 oracle (value, gradient, metric): fp64 %.2e %.2e %.2e | fp32 %.2e %.2e %.2e"
          % (errs[torch.float64] + errs[torch.float32]))
    assert max(errs[torch.float64]) < 1e-9
    # fp32 fields, fp64 accumulators: value (an fp64 sum over 1.7e7 fp32 residuals) to ~1e-7, gradient / metric
    # application to a few 1e-6 of their largest entry
    e_val, e_grad, e_met = errs[torch.float32]
    assert e_val < 1e-8 and e_grad < 3e-4 and e_met < 1e-6


@pytest.mark.timeout(600)
def test_config2_full_size_against_the_oracle():
    """BASELINE config 2 at its real size: 2048^2 RGSpace, Poissonian likelihood on exp(correlated field), fp64 -- value,
    gradient, metric application and one mirrored MGVI sample pair against the oracle."""
    from nifty_amd import random
    from nifty_amd.engine import FusedModel, LatentVec, draw_samples
    from nifty_amd.minimization import AbsDeltaEnergyController

    shape = (2048, 2048)
    cores = os.cpu_count() or 1
    cf = orc.CFModel(shape, None, orc.CFParams(offset_mean=2.0), workers=cores)
    rng = np.random.default_rng(23)
    x = {k: 0.1 * a for k, a in cf.draw_latent(rng).items()}
    v = cf.draw_latent(rng)
    data = rng.poisson(np.exp(cf.forward(cf.draw_latent(rng)))).astype(np.int64)
    lh = orc.Likelihood("poisson", data, nonlin="exp")
    model = FusedModel(shape, offset_mean=2.0, likelihood="poisson", nonlin="exp", data=data, dtype=torch.float64, device="cuda:0")
    lin = orc.Linearized(cf, lh, x)
    e_val, e_grad, e_met = _errors(model, lin, x, v, "2048^2 fp64 Poisson")
    print(f"2048^2 fp64 Poisson vs oracle: value {e_val:.2e} gradient {e_grad:.2e} metric {e_met:.2e}")
    assert e_val < 1e-11 and e_grad < 1e-9 and e_met < 1e-9
    random.push_sseq_from_seed(6)
    try:
        res, negs, n_total = draw_samples(model, LatentVec.from_dict(model, x), 1, True,
                                          lambda: AbsDeltaEnergyController(0.05, iteration_limit=4))
    finally:
        random.pop_sseq()
    ores, onegs = orc.draw_samples(cf, lh, x, 1, True, np.random.SeedSequence(6),
                                   lambda: orc.AbsDeltaEnergyController(0.05, iteration_limit=4))
    assert n_total == 2 and negs == onegs == [False, True]
    e_s = gl.lat_relerr(res[0].to_dict(), ores[0])
    print(f"2048^2 fp64 Poisson MGVI sample vs oracle: {e_s:.2e}")
    assert e_s < 1e-7
