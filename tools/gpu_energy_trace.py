"""KL energy per MGVI iteration of the bench workload at a reduced size (convergence trace).
usage: python tools/gpu_energy_trace.py [n] [iterations] [numpy|device]"""
import math, sys
import numpy as np, torch
sys.path.insert(0, ".")
from nifty_amd import random
from nifty_amd.engine import FusedModel, mgvi_iteration
from nifty_amd.minimization import AbsDeltaEnergyController, NewtonCG

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
its = int(sys.argv[2]) if len(sys.argv) > 2 else 12
mode = sys.argv[3] if len(sys.argv) > 3 else "numpy"
dev = torch.device("cuda:0")
shape = (n, n, n)
model = FusedModel(shape, offset_mean=2.0, offset_std=(1e-1, 3e-2), fluctuations=(1.0, 5e-1), loglogavgslope=(-3.0, 2e-1),
                   flexibility=(1.0, 2e-1), asperity=(5e-1, 5e-2), likelihood="gaussian", icov=100.0, dtype=torch.float32, device=dev)
random.push_sseq_from_seed(42)
gen = torch.Generator(device=dev).manual_seed(42) if mode == "device" else None
truth = model.draw_prior(gen)
print("truth hyper-parameter latents", truth.small[:5].cpu().numpy())
data = model.signal(truth)
print("signal mean %.3f std %.3f" % (float(data.mean()), float(data.std())))
noise = torch.randn(shape, device=dev, generator=gen) * 0.1 if gen is not None else random.Random.normal_on_device(np.float32, shape, 0.0, 0.1, dev)
data.add_(noise)
model.set_data(data, 100.0)
mean = 0.1 * model.draw_prior(gen)
rng = torch.Generator(device=dev).manual_seed(1234) if mode == "device" else None
for i in range(its):
    ic = lambda: AbsDeltaEnergyController(0.05, iteration_limit=20)
    mini = NewtonCG(AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=3), max_cg_iterations=20)
    mean, kl = mgvi_iteration(model, mean, 4, ic, mini, mirror_samples=True, device_rng=rng)
    print(i, "KL energy / N = %.4f" % (kl.value / model.N), " latents", mean.small[:5].cpu().numpy().round(3))
