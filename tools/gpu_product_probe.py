"""A product-spectrum correlated field (time x space: two add_fluctuations calls) on the device through the operator graph with
the fused product node: time of value + gradient + one metric application and of one optimize_kl iteration.
usage: python tools/gpu_product_probe.py [nt] [nx] [ny] [f32|f64]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import nifty_amd as ift

nt, nx, ny = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (256, 128, 64)))
dt = np.float32 if (len(sys.argv) > 4 and sys.argv[4] == "f32") else np.float64
ift.random.push_sseq_from_seed(42)
cfm = ift.CorrelatedFieldMaker("p")
cfm.add_fluctuations(ift.RGSpace((nt,), (0.5,)), (1.0, 5e-1), (1.0, 2e-1), (5e-1, 5e-2), (-3.0, 2e-1), prefix="t")
cfm.add_fluctuations(ift.RGSpace((nx, ny)), (0.7, 3e-1), (1.2, 2e-1), (4e-1, 5e-2), (-2.5, 2e-1), prefix="s")
cfm.set_amplitude_total_offset(2.0, (1e-1, 3e-2))
cf = cfm.finalize()
truth = ift.from_random(cf.domain, dtype=dt, device_id=0)
d = cf(truth) + 0.1 * ift.from_random(cf.target, dtype=dt, device_id=0)
lh = ift.GaussianEnergy(d, ift.ScalingOperator(cf.target, 100.0, dt)) @ cf
ham = ift.StandardHamiltonian(lh, ift.AbsDeltaEnergyController(0.05, iteration_limit=5), prior_sampling_dtype=dt)
x = 0.1 * ift.from_random(cf.domain, dtype=dt, device_id=0)
v = ift.from_random(cf.domain, dtype=dt, device_id=0)


def evaluation():
    lin = ham(ift.Linearization.make_var(x, want_metric=True))
    return lin, lin.gradient, lin.metric(v)


def timed(fn, reps):
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


ms, (lin, g, mv) = timed(evaluation, 5)
print(f"{nt} x {nx} x {ny} {np.dtype(dt).name}: value + gradient + one metric application: {ms:.2f} ms "
      f"(value {float(lin.val.asnumpy()):.10e})")
lin = ham(ift.Linearization.make_var(x, want_metric=True))
ms, _ = timed(lambda: lin.metric(v), 20)
print(f"  one metric application alone: {ms:.2f} ms")
minimizer = ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, iteration_limit=3, convergence_level=2), max_cg_iterations=20)


def iteration():
    with ift.random.Context(7):
        return ift.optimize_kl(lh, 1, 4, minimizer, ift.AbsDeltaEnergyController(0.05, iteration_limit=20),
                               initial_position=x, device_id=0, plot_energy_history=False, plot_minisanity_history=False)


ms, sl = timed(iteration, 2)
print(f"  one optimize_kl iteration (4 samples, 20-step sampling CG, 3 Newton steps): {ms:.1f} ms")
