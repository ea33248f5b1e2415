"""Do two independent transforms on two streams overlap their load / compute / store phases?  (1024^3 fp32 plain Hartley)"""
import sys, time
import torch
sys.path.insert(0, ".")
from nifty_amd import backend as B, _lib as L

shape = (1024, 1024, 1024)
dev = torch.device("cuda:0")
x1 = torch.randn(shape, dtype=torch.float32, device=dev)
x2 = torch.randn(shape, dtype=torch.float32, device=dev)
o1, o2 = torch.empty_like(x1), torch.empty_like(x2)
p1, p2 = B.Plan(shape, torch.float32, 1, dev), B.Plan(shape, torch.float32, 1, dev)
lib = L.load()


def run(plan, x, o):
    L.check(lib.nk_hartley(plan.handle, x.data_ptr(), o.data_ptr(), 1.0, 0, plan.workspace.data_ptr(), B._stream()))


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        run(p1, x1, o1)
        run(p2, x2, o2)
    torch.cuda.synchronize()
    seq = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(10):
        with torch.cuda.stream(s1):
            run(p1, x1, o1)
        with torch.cuda.stream(s2):
            run(p2, x2, o2)
    torch.cuda.synchronize()
    par = (time.perf_counter() - t0) / 20
    print(f"per transform: sequential {seq*1e3:.3f} ms, two streams {par*1e3:.3f} ms")
