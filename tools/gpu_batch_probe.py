"""Host enqueue time vs device time of the batched launch sets (nifty_amd/batched.py) on BASELINE config 2's model.
usage: python tools/gpu_batch_probe.py [n=2048] [samples=8]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from nifty_amd import batched, random
from nifty_amd.engine import FusedKL, FusedModel, draw_samples
from nifty_amd.minimization import AbsDeltaEnergyController

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
model = FusedModel((n, n), offset_mean=2.0, likelihood="poisson", nonlin="exp", device="cuda:0")
random.push_sseq_from_seed(42)
truth = model.draw_prior()
model.set_data(torch.poisson(model.signal(truth).double()).to(torch.int64))
mean = 0.1 * model.draw_prior()
res, negs, nt = draw_samples(model, mean, S // 2, True, lambda: AbsDeltaEnergyController(0.05, iteration_limit=3))
kl = FusedKL(model, mean, res, negs, nt)
d = 0.5 * mean


def measure(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    host = 0.0
    e0.record()
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); host += time.perf_counter() - t0
    e1.record(); torch.cuda.synchronize()
    return 1e3 * host / reps, e0.elapsed_time(e1) / reps


print("batched ready:", batched.ready(model), "samples", len(res))
h, g = measure(lambda: kl.apply_metric(d))
print(f"KL metric application, {len(res)} samples: host enqueue {h:.3f} ms, device (back to back) {g:.3f} ms")
h, g = measure(lambda: kl.at(mean), reps=5)
print(f"KL value/gradient, {len(res)} samples: host enqueue {h:.3f} ms, device {g:.3f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    kl.apply_metric(d)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
