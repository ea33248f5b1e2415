"""Device-copy ceiling of the box: read-only, write-only and copy streams over a latent-sized array,
through libniftyk's own vector kernels and through torch (hipMemcpyAsync D2D / fill).  usage: python tools/gpu_copy_ceiling.py [n] [f32|f64]"""
import sys
import torch
sys.path.insert(0, ".")
from nifty_amd import backend as B

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
dt = torch.float32 if (len(sys.argv) < 3 or sys.argv[2] == "f32") else torch.float64
dev = torch.device("cuda:0")
x, y = torch.randn(n, dtype=dt, device=dev), torch.empty(n, dtype=dt, device=dev)
bs = x.element_size()


def timed(tag, streams, fn, reps=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{tag:28s} {ms:7.3f} ms  {streams * n * bs / ms / 1e6:7.1f} GB/s")


timed("nk_vdot(x,x)  read 1", 1, lambda: B.vdot(x, x))
timed("nk_sum        read 1", 1, lambda: B.vsum(x))
timed("nk_axpby copy 1R+1W", 2, lambda: B.axpby(1.0, x, out=y))
timed("torch copy_   1R+1W", 2, lambda: y.copy_(x))
timed("torch fill_   1W", 1, lambda: y.fill_(1.0))
timed("nk_axpby 2R+1W", 3, lambda: B.axpby(1.0, x, 0.5, y, out=y))
