"""Runs the engine-vs-oracle parity check of tests/test_engine_gpu.py over a randomised sweep of grid shapes (fast
power-of-two lengths 64…4096 mixed with generic mixed-radix lengths), both field types.  A development probe: the
committed test suite holds the fixed cases, this looks for size regimes nobody thought of.
usage: python tools/gpu_shape_sweep.py [n_shapes] [seed]"""
import sys
import time
import traceback

import numpy as np
import torch

sys.path.insert(0, ".")
from tests.test_engine_gpu import test_engine_vs_oracle_seeded as check  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
FAST = [64, 128, 256, 512, 1024, 2048, 4096]
ODD = [6, 10, 12, 30, 48, 96, 100, 120, 192, 250, 320, 384, 640, 1000]
KINDS = [("gaussian", None), ("poisson", "exp"), ("gaussian", "sigmoid")]
bad = 0
for it in range(n):
    ndim = int(rng.choice([1, 2, 3], p=[0.1, 0.35, 0.55]))
    while True:
        shape = tuple(int(rng.choice(FAST if rng.random() < 0.7 else ODD)) for _ in range(ndim))
        if np.prod(shape) <= (1 << 24) and shape[-1] % 2 == 0:
            break
    kind, nonlin = KINDS[int(rng.integers(len(KINDS)))]
    for dtype in (torch.float64, torch.float32):
        t0 = time.time()
        try:
            check.__wrapped__(shape, kind, nonlin, dtype) if hasattr(check, "__wrapped__") else check(shape, kind, nonlin, dtype)
            res = "ok"
        except Exception as e:  # noqa: BLE001
            bad += 1
            res = "FAIL " + "".join(traceback.format_exception_only(type(e), e)).strip()[:300]
        print(f"{str(shape):20s} {kind:9s} {str(nonlin):8s} {str(dtype):14s} {time.time() - t0:6.1f}s  {res}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
