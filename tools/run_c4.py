"""BASELINE config 4 on one GPU through the nifty.cl-shaped API: 2-D RGSpace CorrelatedField, sigmoid, masked
LOSResponse (n_los random lines, demos/cl/getting_started_3.py:48-51,98-100), Gaussian noise 1e-3, geoVI.
Usage: python tools/run_c4.py [n] [n_los] [iterations]   (defaults 4096 10000 1)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import nifty_amd as ift

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n_los = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ift.random.push_sseq_from_seed(42)
sp = ift.RGSpace((n, n))
cfm = ift.CorrelatedFieldMaker("")
cfm.add_fluctuations(sp, (1.0, 5e-1), (1.0, 2e-1), (5e-1, 5e-2), (-3.0, 2e-1))
cfm.set_amplitude_total_offset(2.0, (1e-1, 3e-2))
cf = cfm.finalize()
rng = np.random.default_rng(1)
t0 = time.perf_counter()
R = ift.LOSResponse(sp, rng.uniform(size=(2, n_los)), rng.uniform(size=(2, n_los)))
print(f"LOS set-up {time.perf_counter() - t0:.1f} s, nnz {len(R._col)}")
flags = np.zeros(n_los, dtype=bool); flags[rng.integers(0, n_los, n_los // 20)] = True
resp = ift.MaskOperator(ift.makeField(R.target, flags)) @ R @ cf.ptw("sigmoid")
truth = ift.from_random(cf.domain, device_id=0)
d = resp(truth) + ift.from_random(resp.target, device_id=0) * np.sqrt(1e-3)
lh = ift.GaussianEnergy(d, ift.ScalingOperator(resp.target, 1e3, np.float64)) @ resp
ic = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=20)
mk = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=3), max_cg_iterations=20)
nl = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, iteration_limit=5, convergence_level=2))
import torch
torch.cuda.synchronize(); t0 = time.perf_counter()
sl, mean = ift.optimize_kl(lh, iters, 4, mk, ic, nonlinear_sampling_minimizer=nl, output_directory=None,
                           return_final_position=True, device_id=0)
torch.cuda.synchronize()
print(f"C4 {n}x{n}, n_los {n_los}, geoVI 4 samples: {(time.perf_counter() - t0) / iters:.1f} s / iteration")
