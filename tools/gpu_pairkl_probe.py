"""One KL metric application with two samples (one pair launch of the final pass) at 1024^3 fp32: time per application.
For A/B runs of library variants: NK_LIB_PATH=build/libniftyk_<tag>.so python tools/gpu_pairkl_probe.py [edge]"""
import sys
import torch
sys.path.insert(0, ".")
from nifty_amd import random
from nifty_amd.engine import FusedKL, FusedModel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = FusedModel((n, n, n), offset_mean=2.0, likelihood="gaussian", icov=100.0, dtype=torch.float32, device="cuda:0")
random.push_sseq_from_seed(3)
model.set_data(model.signal(model.draw_prior()), 100.0)
mean = 0.1 * model.draw_prior()
res = 0.01 * model.draw_prior()
d = model.draw_prior()
import os
ns = int(os.environ.get('NK_PROBE_SAMPLES', '2'))
kl = FusedKL(model, mean, [res] * ns, [i % 2 == 1 for i in range(ns)], ns)
for _ in range(2):
    kl.apply_metric(d)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    kl.apply_metric(d)
e1.record()
torch.cuda.synchronize()
torch.cuda.synchronize()
print(f"KL metric application, {ns} samples: {e0.elapsed_time(e1) / 10:.3f} ms")
