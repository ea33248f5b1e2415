"""Two samples of the KL metric: two nk_hartley_sandwich calls (read-modify-write of the output per sample) against one
nk_hartley_sandwich_pair.  usage: python tools/gpu_pair_probe.py [n]"""
import ctypes, sys
import torch
sys.path.insert(0, ".")
from nifty_amd import _lib as L, backend as B
from nifty_amd.engine import FusedModel
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
shape = (n, n, n)
model = FusedModel(shape, offset_mean=2.0, likelihood="gaussian", icov=100.0, dtype=torch.float32, device="cuda:0")
gen = torch.Generator(device="cuda").manual_seed(1)
xs = [0.1 * model.draw_prior(gen) for _ in range(2)]
model.set_data(model.signal(xs[0]), 100.0)
lps = [model.linearize(x) for x in xs]
d = model.draw_prior(gen)
lib = L.load()
ws2 = torch.empty_like(model.plan.workspace)
daf = [torch.empty_like(model.dafield) for _ in range(2)]
w8 = [torch.empty_like(model.w8) for _ in range(2)]
out = torch.zeros_like(d.xi)
dot = torch.zeros(1, dtype=torch.float64, device="cuda")


def fuse(i, pair):
    lp = lps[i]
    f = model._fuse()
    f.pro, f.in_, f.in2 = L.PRO_AMP_JVP, d.xi.data_ptr(), lp.x.xi.data_ptr()
    f.pidx, f.amp, f.damp = model.pidx.data_ptr(), lp.amp.data_ptr(), model.damp.data_ptr()
    f.afield, f.dafield = lp.afield.data_ptr(), daf[i].data_ptr()
    f.mul_scalar = lp.mid_scalar
    f.epi, f.out, f.scale = L.EPI_VJP, out.data_ptr(), model.h_dvol * 0.125
    f.xi, f.abar, f.w8 = lp.x.xi.data_ptr(), model.abar.data_ptr(), w8[i].data_ptr()
    if pair:
        if i == 0:
            f.accumulate = 1
        else:
            f.addend, f.addend_scale, f.value = d.xi.data_ptr(), 1.0, dot.data_ptr()
    else:
        f.accumulate = 1
        if i == 1:
            f.addend, f.addend_scale, f.value = d.xi.data_ptr(), 1.0, dot.data_ptr()
    return f


for i in range(2):
    model.damp.copy_(torch.randn_like(model.damp) * 0.01)
    model._amp_field(model.damp, out=daf[i])


def singles():
    for i in range(2):
        f = fuse(i, False)
        B.hartley_sandwich(model.plan, f, model.h_dvol)


def pair():
    fa, fb = fuse(0, True), fuse(1, True)
    L.check(lib.nk_hartley_sandwich_pair(model.plan.handle, ctypes.byref(fa), ctypes.byref(fb), model.h_dvol, 0,
                                         model.plan.workspace.data_ptr(), ws2.data_ptr(), B._stream()))


def timed(tag, fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{tag:30s} {e0.elapsed_time(e1) / reps:8.3f} ms")


out.zero_(); dot.zero_(); singles(); ref, dref, wref = out.clone(), dot.clone(), [w.clone() for w in w8]
out.zero_(); dot.zero_(); pair()
print("pair vs singles: out", float((out - ref).abs().max() / ref.abs().max()), "dot", float(((dot - dref) / dref).abs()),
      "w8", [float((a - b).abs().max() / b.abs().max()) for a, b in zip(w8, wref)])
timed("two sandwiches", singles)
timed("sandwich pair", pair)
lib.nk_profile_enable(1); bench.collect_profile()
for _ in range(3):
    pair()
prof = bench.collect_profile()
for (k, p, e), (ms, c) in sorted(prof.items()):
    print(f"  pair  {bench.KERNEL_NAMES[k]:8s} avg {ms / c:7.3f} ms x{c}")
for _ in range(3):
    singles()
prof = bench.collect_profile()
for (k, p, e), (ms, c) in sorted(prof.items()):
    print(f"  single {bench.KERNEL_NAMES[k]:8s} avg {ms / c:7.3f} ms x{c}")
