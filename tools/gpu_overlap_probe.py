"""Does a latency-bound kernel (the octant gather of da, the shell-binned scatter) overlap with the bandwidth-bound
transform passes when it runs on a second stream?  usage: python tools/gpu_overlap_probe.py"""
import ctypes, sys
import torch
sys.path.insert(0, ".")
from nifty_amd import _lib as L, backend as B
from nifty_amd.engine import FusedModel

shape = (1024, 1024, 1024)
model = FusedModel(shape, offset_mean=0.0, likelihood="gaussian", icov=1.0, dtype=torch.float32, device="cuda:0")
x = torch.randn(shape, dtype=torch.float32, device="cuda")
table = torch.randn(model.nb, dtype=torch.float32, device="cuda")
out8 = torch.empty(model.field_shape, dtype=torch.float32, device="cuda")
w8 = torch.rand(model.w8.shape, dtype=torch.float64, device="cuda")
abar = torch.zeros(model.nb, dtype=torch.float64, device="cuda")
shp = (ctypes.c_int64 * 3)(*shape)
lib = L.load()
side = torch.cuda.Stream()


def gather():
    B.gather(table, model.pidx8, model.field_shape, out=out8)


def scatter():
    L.check(lib.nk_octant_scatter_k2(3, shp, w8.data_ptr(), model.pidx.data_ptr(), model.bin_k2.data_ptr(), model.nb,
                                     model.scatter_scratch.data_ptr(), abar.data_ptr(), None, B._stream()))


def timed(tag, fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{tag:44s} {e0.elapsed_time(e1) / reps:7.3f} ms")


def concurrent(small):
    def run():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            small()
        B.hartley(x)
        torch.cuda.current_stream().wait_stream(side)
    return run


timed("hartley (3 passes)", lambda: B.hartley(x))
timed("gather", gather)
timed("scatter_k2", scatter)
timed("gather then hartley, one stream", lambda: (gather(), B.hartley(x)))
timed("gather on a side stream || hartley", concurrent(gather))
timed("scatter then hartley, one stream", lambda: (scatter(), B.hartley(x)))
timed("scatter on a side stream || hartley", concurrent(scatter))
