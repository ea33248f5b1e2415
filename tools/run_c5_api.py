"""The headline workload through the USER-LEVEL API (optimize_kl with the fusion pass), not through bench.py's direct
engine calls: 3-D RGSpace CorrelatedField + Gaussian likelihood, fp32 fields, 4 mirrored sample pairs.
Usage: python tools/run_c5_api.py [n] [iterations] [sampling_rng]   (defaults 1024 2 numpy)"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import nifty_amd as ift

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if len(sys.argv) > 3:  # third argument: sampling_rng mode (default: the reference's numpy streams, computed on the device)
    ift.config.update("sampling_rng", sys.argv[3])
ift.random.push_sseq_from_seed(42)
sp = ift.RGSpace((n, n, n))
cfm = ift.CorrelatedFieldMaker("")
cfm.add_fluctuations(sp, (1.0, 5e-1), (1.0, 2e-1), (5e-1, 5e-2), (-3.0, 2e-1))
cfm.set_amplitude_total_offset(2.0, (1e-1, 3e-2))
cf = cfm.finalize()
d = ift.from_random(cf.target, dtype=np.float32, device_id=0, mean=2.0, std=0.1)
lh = ift.GaussianEnergy(d, ift.ScalingOperator(cf.target, 100.0, np.float32)) @ cf
x0 = 0.1 * ift.from_random(cf.domain, dtype=np.float32, device_id=0)
ic = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=20)
mk = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=3), max_cg_iterations=20)
torch.cuda.synchronize(); t0 = time.perf_counter()
sl, mean = ift.optimize_kl(lh, iters, 4, mk, ic, output_directory=None, return_final_position=True, initial_position=x0,
                           device_id=0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"optimize_kl {n}^3 fp32, 8 samples: {dt:.2f} s / iteration; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
