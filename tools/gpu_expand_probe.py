"""Octant expansion of a bin table (da[pidx] per metric application): plain gather through the int32 bin index of the
octant points vs nk_octant_expand_k2 (dense k^2 table, no index stream).  Usage: python tools/gpu_expand_probe.py [n] [f32|f64]"""
import sys

import torch

sys.path.insert(0, ".")
from nifty_amd.engine import FusedModel  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dt = torch.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else torch.float32
model = FusedModel((n, n, n), offset_mean=2.0, likelihood="gaussian", icov=100.0, dtype=dt, device="cuda:0")
table = torch.randn(model.nb, dtype=torch.float64, device="cuda")


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


assert model.k2_dense is not None
new = model._amp_field(table).clone()
t_new = timed(lambda: model._amp_field(table, out=model.dafield))
dense, model.k2_dense = model.k2_dense, None
old = model._amp_field(table).clone()
t_old = timed(lambda: model._amp_field(table, out=model.dafield))
model.k2_dense = dense
import os
print(f"{n}^3 {dt}: gather through pidx8 {t_old:.3f} ms, dense k^2 table (NK_EXPAND_SHELL={os.environ.get('NK_EXPAND_SHELL', '1')}) "
      f"{t_new:.3f} ms, identical: {torch.equal(old, new)}")
