import sys, torch
sys.path.insert(0, ".")
from nifty_amd import backend as B
shape = tuple(int(s) for s in sys.argv[1].split(","))
dt = torch.float64 if sys.argv[2] == "f64" else torch.float32
x = torch.randn(shape, dtype=dt, device="cuda")
y = B.hartley(x); torch.cuda.synchronize()
F = torch.fft.fftn(x.double()); ref = F.real + F.imag
print(shape, dt, "err", ((y.double() - ref).abs().max() / ref.abs().max()).item(), flush=True)
