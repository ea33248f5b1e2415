"""Value / gradient of the fused engine against the oracle for a few shapes (debugging aid for the likelihood epilogue)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import nifty_oracle as orc
from nifty_amd.engine import FusedModel, LatentVec
from tests import goldenlib as gl

cases = [((64, 64, 64), "gaussian", None), ((64, 4096, 64), "poisson", "exp"), ((64, 64, 1024), "gaussian", None),
         ((4096, 1024), "gaussian", "sigmoid"), ((64, 64), "gaussian", None), ((2048, 64, 64), "gaussian", None)]
for shape, kind, nonlin in cases:
    for dtype in (torch.float64,):
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
        lin = orc.Linearized(cf, lh, x)
        val, grad = lin.value_grad()
        lp = model.linearize(LatentVec.from_dict(model, x))
        print(shape, kind, nonlin, "value relerr", abs(float(lp.value.item()) - val) / abs(val), "grad relerr",
              gl.lat_relerr(lp.grad.to_dict(), grad), flush=True)
