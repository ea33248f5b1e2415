"""Times every fused prologue/epilogue variant of the transform separately (per pass kernel, HIP events)."""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, ".")
from nifty_amd import _lib as L, backend as B
from nifty_amd.engine import FusedModel, LatentVec
import bench

shape = tuple(int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "1024,1024,1024").split(","))
dtype = torch.float32 if (len(sys.argv) < 3 or sys.argv[2] == "f32") else torch.float64
dev = torch.device("cuda:0")
model = FusedModel(shape, offset_mean=2.0, likelihood="gaussian", icov=100.0, dtype=dtype, device=dev)
gen = torch.Generator(device=dev).manual_seed(1)
x = 0.1 * model.draw_prior(gen)
data = model.signal(x); model.set_data(data, 100.0)
d = model.draw_prior(gen)
lib = L.load()
N = int(np.prod(shape)); b = 4 if dtype == torch.float32 else 8

def report(tag):
    prof = bench.collect_profile()
    for (k, p, e), (ms, c) in sorted(prof.items()):
        ab = bench.algorithmic_bytes(k, p, e, N, b, model.const_mid)
        print(f"  {tag:12s} {bench.KERNEL_NAMES[k]:8s} pro={bench.PRO_NAMES[p]:8s} epi={bench.EPI_NAMES[e]:10s} avg {ms/c:8.3f} ms  x{c}  {ab/ (ms/c*1e-3)/1e9 if ms>0 else 0:8.1f} GB/s")

lp = model.linearize(x)
lib.nk_profile_enable(1); bench.collect_profile()
for _ in range(3):
    out = B.hartley(d.xi)
report("plain")
for _ in range(3):
    lp = model.linearize(x)
report("linearize")
for _ in range(3):
    q = model.metric(lp, d)
report("metric")
acc_out = LatentVec(torch.zeros_like(d.xi), None)
dot = torch.zeros(1, dtype=torch.float64, device=dev)
for _ in range(3):
    model.lh_metric_accumulate(lp, d, acc_out, 0.125, True)
report("metric 1st")
for _ in range(3):
    model.lh_metric_accumulate(lp, d, acc_out, 0.125, False)
report("metric acc")
for _ in range(3):
    model.lh_metric_accumulate(lp, d, acc_out, 0.125, False, identity=1.0, dot_out=dot)
report("metric acc+d")
for _ in range(3):
    s = model.signal(x)
report("signal")
if B.plan_sandwich(model.plan):
    out = torch.empty_like(d.xi)
    for _ in range(3):
        f = L.Fuse()
        f.pro, f.in_, f.epi, f.out, f.scale, f.mul_scalar = L.PRO_PLAIN, d.xi.data_ptr(), L.EPI_AFFINE, out.data_ptr(), 1.0 / N, 1.0
        B.hartley_sandwich(model.plan, f, 1.0)
    report("sandwich")
    mid = torch.rand_like(d.xi)
    for _ in range(3):
        f = L.Fuse()
        f.pro, f.in_, f.epi, f.out, f.scale, f.mul_scalar = L.PRO_PLAIN, d.xi.data_ptr(), L.EPI_AFFINE, out.data_ptr(), 1.0 / N, 1.0
        f.mul = mid.data_ptr()
        B.hartley_sandwich(model.plan, f, 1.0)
    report("sandwich+mid")
