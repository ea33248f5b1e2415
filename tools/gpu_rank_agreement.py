"""One MGVI iteration of a small fp64 model with 1 or several ranks (gloo, all on GPU 0): prints checksums of the samples, of
the KL value / gradient at the start and of the result, for comparing rank counts.
  python tools/gpu_rank_agreement.py                       # one process
  NK_DIST_BACKEND=gloo NK_SHARE_DEVICE=1 python -m torch.distributed.run --nproc-per-node 2 tools/gpu_rank_agreement.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from nifty_amd import parallel, random  # noqa: E402
from nifty_amd.engine import FusedKL, FusedModel, draw_samples  # noqa: E402
from nifty_amd.minimization import AbsDeltaEnergyController, NewtonCG  # noqa: E402

comm = None
if "WORLD_SIZE" in os.environ:
    comm, _ = parallel.init(os.environ.get("NK_DIST_BACKEND", "gloo"))
rank = 0 if comm is None else comm.rank
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
shape = (64, 64, 64)
model = FusedModel(shape, offset_mean=2.0, likelihood="gaussian", icov=100.0, dtype=torch.float64, device=dev)
random.push_sseq_from_seed(42)
truth = model.draw_prior()
data = model.signal(truth)
data.add_(random.Random.normal_on_device(np.float64, shape, 0.0, 0.1, dev))
model.set_data(data, 100.0)
mean = 0.1 * model.draw_prior()
ic = lambda: AbsDeltaEnergyController(0.05, iteration_limit=20)  # noqa: E731
res, negs, n_total = draw_samples(model, mean, 4, True, ic, comm)
plan = parallel.SamplePlan(4, True, comm)
for r, neg in zip(res, negs):
    print(f"rank {rank}: sample neg={neg} sum {float(r.xi.sum()):.15e} sumsq {float((r.xi ** 2).sum()):.15e} "
          f"small {float(r.small.sum()):.15e}", flush=True)
kl = FusedKL(model, mean, res, negs, n_total, comm)
if rank == 0:
    print(f"KL at start: value {kl.value!r} grad sumsq {float((kl.gradient.xi ** 2).sum()):.15e}", flush=True)
mini = NewtonCG(AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=3), max_cg_iterations=20)
with parallel.lockstep(comm):
    kl2, _ = mini(kl)
if rank == 0:
    print(f"KL after minimisation: value {kl2.value!r} position sumsq {float((kl2.position.xi ** 2).sum()):.15e}", flush=True)
# further iterations exactly like bench.py's step()
from nifty_amd.engine import mgvi_iteration  # noqa: E402

mean = kl2.position
from nifty_amd import minimization  # noqa: E402

_orig = model.draw_mgvi_sample


def _counted(lp, controller, device_rng=None):
    before = minimization.counters["cg_iterations"]
    out = _orig(lp, controller, device_rng)
    y = out[1]
    print(f"rank {rank}: linear sample with {minimization.counters['cg_iterations'] - before} CG iterations, "
          f"sumsq {float((y.xi ** 2).sum()):.15e}, b sumsq {float((out[0].xi ** 2).sum()):.15e}", flush=True)
    return out


model.draw_mgvi_sample = _counted
for it in range(1):
    mini = NewtonCG(AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=3), max_cg_iterations=20)
    mean, klit = mgvi_iteration(model, mean, 4, ic, mini, mirror_samples=True, comm=comm)
    for r, neg in zip(klit.residuals, klit.negs):
        print(f"iteration {it + 2} rank {rank}: sample neg={neg} sumsq {float((r.xi ** 2).sum()):.15e}", flush=True)
    if rank == 0:
        print(f"iteration {it + 2}: KL {klit.value!r} position sumsq {float((mean.xi ** 2).sum()):.15e}", flush=True)
