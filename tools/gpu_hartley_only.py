"""Runs a few plain Hartley transforms (for rocprofv3 counter collection)."""
import sys, torch
sys.path.insert(0, ".")
from nifty_amd import backend as B
shape = tuple(int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "1024,1024,1024").split(","))
dtype = torch.float32 if (len(sys.argv) < 3 or sys.argv[2] == "f32") else torch.float64
x = torch.randn(shape, dtype=dtype, device="cuda")
out = torch.empty_like(x)
for _ in range(3):
    B.hartley(x, out=out)
torch.cuda.synchronize()
print("done")
