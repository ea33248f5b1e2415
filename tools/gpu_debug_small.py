import sys, torch, numpy as np
sys.path.insert(0, ".")
from nifty_amd import backend as B
for shape in [(64, 64), (128, 64), (64, 64, 64), (256,), (64, 128, 256)]:
    for dt in (torch.float64, torch.float32):
        x = torch.randn(shape, dtype=dt, device="cuda")
        print("run", shape, dt, flush=True)
        y = B.hartley(x)
        torch.cuda.synchronize()
        F = torch.fft.fftn(x.double())
        ref = F.real + F.imag
        print("  err", ((y.double() - ref).abs().max() / ref.abs().max()).item(), flush=True)
print("plain ok", flush=True)
from nifty_amd.engine import FusedModel, LatentVec
m = FusedModel((64, 64, 64), offset_mean=1.0, likelihood="gaussian", icov=10.0, device="cuda:0")
g = torch.Generator(device="cuda").manual_seed(1)
x = 0.1 * m.draw_prior(g)
d = m.signal(x); torch.cuda.synchronize(); print("signal ok", flush=True)
m.set_data(d, 10.0)
lp = m.linearize(x); torch.cuda.synchronize(); print("linearize ok", float(lp.value.item()), flush=True)
q = m.metric(lp, m.draw_prior(g)); torch.cuda.synchronize(); print("metric ok", flush=True)
