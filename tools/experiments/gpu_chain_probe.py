"""Cache-chained sandwich (NK_SANDWICH_CHAIN = slabs per chunk): time of one metric application and bit-identity against
the unchained passes.  Run once per setting (the library reads the variable once):
    for w in 0 16 32 64; do NK_SANDWICH_CHAIN=$w python tools/gpu_chain_probe.py; done"""
import os
import sys

import torch

sys.path.insert(0, ".")
from nifty_amd import random  # noqa: E402
from nifty_amd.engine import FusedModel  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dt = torch.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else torch.float32
model = FusedModel((n, n, n), offset_mean=2.0, likelihood="gaussian", icov=100.0, dtype=dt, device="cuda:0")
random.push_sseq_from_seed(3)
x = model.draw_prior() * 0.1
model.set_data(model.signal(model.draw_prior()), 100.0)
d = model.draw_prior()
lp = model.linearize(x)
q = model.metric(lp, d)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 10
e0.record()
for _ in range(reps):
    q = model.metric(lp, d)
e1.record()
torch.cuda.synchronize()
print(f"NK_SANDWICH_CHAIN={os.environ.get('NK_SANDWICH_CHAIN', '0')}: {n}^3 {dt} metric application {e0.elapsed_time(e1) / reps:.3f} ms; "
      f"checksum {float(q.xi.double().sum()):.17g} {float(q.xi.double().abs().max()):.17g} {float(q.small.sum()):.17g}")
