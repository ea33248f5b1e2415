"""The two sparse products of BASELINE config 4 in isolation (4096^2 grid, 1e4 random lines): the grid-tiled TIMES
(nk_tiled_rowsum) for several tile shapes against the wavefront-per-row kernel, the staged ADJOINT_TIMES against the
thread-per-row kernel (NK_ROWSUM_STAGED=0 in the environment), live HIP events."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from nifty_amd import backend as B
from nifty_amd.los_response import SparseResponse, los_matrix, tiled_plan

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n_los = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
shape = (n, n)
rng = np.random.default_rng(1)
t0 = time.time()
rowptr, col, wgt = los_matrix(shape, (1.0 / n, 1.0 / n), rng.uniform(size=(2, n_los)), rng.uniform(size=(2, n_los)))
print(f"matrix: {len(col)} entries, set-up {time.time() - t0:.1f} s")
dev = torch.device("cuda:0")
x = torch.randn(shape, dtype=torch.float64, device=dev)
y = torch.randn(n_los, dtype=torch.float64, device=dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


os.environ["NK_TILED_RESPONSE"] = "0"
sr = SparseResponse(rowptr, col, wgt, n * n, shape)
ref = sr.times(x)
print(f"TIMES   wavefront per row        {timed(lambda: sr.times(x)):8.1f} us")
print(f"ADJOINT {'staged' if os.environ.get('NK_ROWSUM_STAGED', '1') != '0' else 'thread per row'}   {timed(lambda: sr.adjoint(y)):8.1f} us")
for th, tw in ((32, 64), (64, 64), (32, 32), (16, 64), (64, 32), (16, 128), (128, 32)):
    t0 = time.time()
    plan = tiled_plan(rowptr, col, wgt, shape, th, tw)
    tm = B.TiledMatrix(plan, dev)
    out = torch.empty(n_los, dtype=torch.float64, device=dev)
    tm.rowsum([x.reshape(-1)], [out])
    err = float((out - ref).abs().max() / ref.abs().max())
    us = timed(lambda: tm.rowsum([x.reshape(-1)], [out]))
    nbytes = len(plan["loc"]) * 6.5 + x.numel() * 8 + plan["n_slots"] * 16
    print(f"TIMES   tiles {th:3d} x {tw:3d}  {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s of its own bytes  items {plan['n_items']} "
          f"padded entries x{len(plan['loc']) / len(col):.3f} slots {plan['n_slots']}  rel. diff {err:.1e}  plan {time.time() - t0:.1f} s")
