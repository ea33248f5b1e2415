"""Where the fp32 error of value+gradient comes from (tests/test_large_oracle_gpu.py measures gradient 5e-5 vs metric 1e-7
of the largest entry at 512^3): the same point through the fp64 and the fp32 engine, stage by stage.
Usage: python tools/gpu_fp32_error_probe.py [edge]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from nifty_amd import random  # noqa: E402
from nifty_amd.engine import FusedModel, LatentVec  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
shape = (n, n, n)
m64 = FusedModel(shape, offset_mean=2.0, likelihood="gaussian", icov=100.0, dtype=torch.float64, device="cuda:0")
m32 = FusedModel(shape, offset_mean=2.0, likelihood="gaussian", icov=100.0, dtype=torch.float32, device="cuda:0")
random.push_sseq_from_seed(1)
x64 = m64.draw_prior() * 0.1
data64 = m64.signal(x64) + 0.1 * torch.randn(shape, dtype=torch.float64, device="cuda")
# identical inputs for both engines: everything rounded to fp32 first
x32 = LatentVec(x64.xi.float(), x64.small.clone())
x64 = LatentVec(x32.xi.double(), x64.small.clone())
d32 = data64.float()
m64.set_data(d32.double(), 100.0)
m32.set_data(d32, 100.0)


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def rms(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


s64, s32 = m64.signal(x64), m32.signal(x32)
print(f"signal s = 2 + HT(a xi): max {rel(s32, s64):.2e} rms {rms(s32, s64):.2e}; of the fluctuation part (s - 2): "
      f"max {float((s32.double() - s64).abs().max() / (s64 - 2).abs().max()):.2e}")
l64, l32 = m64.linearize(x64), m32.linearize(x32)
print(f"value {abs(float(l32.value) - float(l64.value)) / abs(float(l64.value)):.2e}")
print(f"grad xi max {rel(l32.grad.xi, l64.grad.xi):.2e} rms {rms(l32.grad.xi, l64.grad.xi):.2e}; "
      f"small max {rel(l32.grad.small, l64.grad.small):.2e}")
# the residual N^-1 (s - d) handed to the adjoint transform (FusedModel.tmp after linearize)
m64.linearize(x64)
g64 = m64.tmp.clone()
m32.linearize(x32)
g32 = m32.tmp.clone()
print(f"dE/ds = icov (s - d): max {rel(g32, g64):.2e} rms {rms(g32, g64):.2e}")
# adjoint transform alone: feed the fp64 residual (rounded) to the fp32 VJP
out64, out32 = LatentVec.zeros(m64), LatentVec.zeros(m32)
m64._vjp(l64, g64, 1.0, None, 0.0, False, out64.xi)
m32._vjp(l32, g64.float(), 1.0, None, 0.0, False, out32.xi)
print(f"VJP of the SAME residual: xi max {rel(out32.xi, out64.xi):.2e} rms {rms(out32.xi, out64.xi):.2e}; "
      f"abar max {rel(m32.abar, m64.abar):.2e}")
m32._vjp(l32, g32, 1.0, None, 0.0, False, out32.xi)
print(f"VJP of the fp32 residual: xi max {rel(out32.xi, out64.xi):.2e} rms {rms(out32.xi, out64.xi):.2e}; "
      f"abar max {rel(m32.abar, m64.abar):.2e}")
# where the largest xi error sits
diff = (out32.xi.double() - out64.xi).abs()
idx = np.unravel_index(int(diff.argmax()), shape)
print("largest xi error at k =", idx, "value there", float(out64.xi[idx]), "max |xi grad|", float(out64.xi.abs().max()))
