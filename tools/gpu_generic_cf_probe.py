"""The generic operator graph (no fusion pass) through CorrelatedFieldOperator on the device: time of one Hamiltonian value +
gradient and of one metric application, with the octant fields (default) and with the per-point table gathers
(NK_CF_OCTANT_FORWARD=0).  usage: python tools/gpu_generic_cf_probe.py [n] [f32|f64]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import nifty_amd as ift

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dt = np.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else np.float64
ift.random.push_sseq_from_seed(42)
sp = ift.RGSpace((n, n, n))
cfm = ift.CorrelatedFieldMaker("")
cfm.add_fluctuations(sp, (1.0, 5e-1), (1.0, 2e-1), (5e-1, 5e-2), (-3.0, 2e-1))
cfm.set_amplitude_total_offset(2.0, (1e-1, 3e-2))
cf = cfm.finalize()
d = ift.from_random(cf.target, dtype=dt, device_id=0, mean=2.0, std=0.1)
lh = ift.GaussianEnergy(d, ift.ScalingOperator(cf.target, 100.0, dt)) @ cf
ham = ift.StandardHamiltonian(lh, ift.AbsDeltaEnergyController(0.05, iteration_limit=5), prior_sampling_dtype=dt)
x = 0.1 * ift.from_random(cf.domain, dtype=dt, device_id=0)
v = ift.from_random(cf.domain, dtype=dt, device_id=0)


def run():
    lin = ham(ift.Linearization.make_var(x, want_metric=True))
    g = lin.gradient
    mv = lin.metric(v)
    return lin, g, mv


for _ in range(2):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    lin, g, mv = run()
torch.cuda.synchronize()
print(f"{n}^3 {np.dtype(dt).name}: value + gradient + one metric application through the generic graph: "
      f"{(time.perf_counter() - t0) / 5 * 1e3:.1f} ms   (value {float(lin.val.asnumpy()):.10e})")
