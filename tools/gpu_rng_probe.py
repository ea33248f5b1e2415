"""Times nk_pcg64_normal (numpy's PCG64 + ziggurat stream on the device) against numpy on the host.
usage: python tools/gpu_rng_probe.py [log2 n]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from nifty_amd import _lib as L, backend as B
import bench

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 1 << lg
dev = torch.device("cuda:0")
for dt in (torch.float32, torch.float64):
    rng = np.random.default_rng(1)
    B.pcg64_normal(rng, 0.0, 1.0, (1 << 20,), dt, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x = B.pcg64_normal(rng, 0.0, 1.0, (n,), dt, dev)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"nk_pcg64_normal 2^{lg} {dt}: {1e3 * (t1 - t0):8.2f} ms wall incl. scratch allocation and the status read-back  "
          f"({n / (t1 - t0) / 1e9:.1f} G normals/s)  mean {float(x.mean()):+.2e} var {float(x.var()):.6f}")
    del x
m = 1 << 26
t0 = time.perf_counter(); y = np.random.default_rng(1).normal(size=m); t1 = time.perf_counter()
print(f"numpy host 2^26 float64: {1e3 * (t1 - t0):8.1f} ms  -> 2^{lg}: {(t1 - t0) * n / m:6.1f} s  (+ upload)")
