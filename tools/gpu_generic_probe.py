"""Plain Hartley transforms on power-of-two and mixed-radix grids: time, and the algorithmic rate (one read + one write of the
array per transformed axis).  usage: python tools/gpu_generic_probe.py"""
import sys
import torch
sys.path.insert(0, ".")
from nifty_amd import backend as B

CASES = [((1024, 1024, 1024), torch.float32), ((768, 768, 768), torch.float32), ((960, 960, 960), torch.float32),
         ((512, 512, 512), torch.float32), ((640, 640, 640), torch.float32), ((384, 384, 384), torch.float32),
         ((2048, 2048), torch.float64), ((1000, 1000), torch.float64), ((1536, 1536), torch.float64), ((3000, 3000), torch.float64),
         ((4096, 4096), torch.float32), ((3072, 3072), torch.float32), ((5000, 5000), torch.float32)]
for shape, dt in CASES:
    x = torch.randn(shape, dtype=dt, device="cuda")
    out = torch.empty_like(x)
    B.hartley(x, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        B.hartley(x, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    byts = len(shape) * 2 * x.numel() * x.element_size()
    print(f"{str(shape):22s} {str(dt):14s} {ms:8.3f} ms  {byts / ms / 1e6:7.1f} GB/s algorithmic", flush=True)
    del x, out
