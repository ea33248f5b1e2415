"""Per-pass timing of the plain Hartley transform for several 3-D shapes (effect of the first-axis stride)."""
import sys, torch
sys.path.insert(0, ".")
from nifty_amd import _lib as L, backend as B
import bench
lib = L.load()
shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1:]] or [(1024, 1024, 1024), (1024, 512, 1024), (1024, 256, 1024), (1024, 64, 1024), (1024, 1024, 256), (512, 1024, 1024), (256, 1024, 1024)]
for shape in shapes:
    x = torch.randn(shape, dtype=torch.float32, device="cuda")
    out = torch.empty_like(x)
    B.hartley(x, out=out)
    lib.nk_profile_enable(1); bench.collect_profile()
    for _ in range(3):
        B.hartley(x, out=out)
    prof = bench.collect_profile()
    N = x.numel()
    line = f"{str(shape):22s} stride {shape[1]*shape[2]*4/1024:8.0f} KiB "
    for (k, p, e), (ms, c) in sorted(prof.items()):
        line += f" {bench.KERNEL_NAMES[k]} {ms/c:6.3f} ms {2*N*4/(ms/c*1e-3)/1e9:6.0f} GB/s |"
    print(line)
    del x, out
