"""2-D octant scatter: shell-binned LDS reduction (nk_octant_scatter_k2) vs plain atomics (nk_octant_scatter).
usage: python tools/gpu_scatter2d_probe.py [n]"""
import ctypes, sys
import torch
sys.path.insert(0, ".")
from nifty_amd import _lib as L, backend as B
from nifty_amd.engine import FusedModel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
shape = (n, n)
model = FusedModel(shape, offset_mean=0.0, likelihood="gaussian", icov=1.0, dtype=torch.float64, device="cuda:0")
w8 = torch.rand(model.w8.shape, dtype=torch.float64, device="cuda")
shp = (ctypes.c_int64 * 2)(*shape)
lib = L.load()


def timed(tag, fn, reps=50):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{tag:28s} {e0.elapsed_time(e1) / reps * 1e3:8.1f} us")


abar1, abar2 = torch.zeros(model.nb, dtype=torch.float64, device="cuda"), torch.zeros(model.nb, dtype=torch.float64, device="cuda")
if model.bin_k2 is not None:
    timed("octant_scatter_k2", lambda: L.check(lib.nk_octant_scatter_k2(2, shp, w8.data_ptr(), model.pidx.data_ptr(), model.bin_k2.data_ptr(),
                                                                     model.nb, model.scatter_scratch.data_ptr(), abar1.data_ptr(), None, B._stream())))


def plain():
    abar2.zero_()
    L.check(lib.nk_octant_scatter(2, shp, w8.data_ptr(), model.pidx.data_ptr(), abar2.data_ptr(), model.merge_swapped, B._stream()))


timed("zero + octant_scatter", plain)
print("agree:", float((abar1 - abar2).abs().max() / abar2.abs().max()))
