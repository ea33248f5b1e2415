"""Bit-identity of the batched launches (NK_BATCH=1) against the stream lanes (NK_BATCH=0) over a randomised sweep of 2-D
grids with a batched twin (512 ... 4096 points per axis), likelihoods, sample counts, field types, MGVI and geoVI: one whole
iteration each way through tests/test_batched_gpu.py's helpers.  usage: python tools/gpu_batch_sweep.py [n] [seed]"""
import os
import sys
import time
import traceback

import numpy as np
import torch

sys.path.insert(0, ".")
from tests import test_batched_gpu as T  # noqa: E402


class _Env:
    def setenv(self, k, v):
        os.environ[k] = v


n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(n):
    shape = tuple(int(rng.choice([512, 1024, 2048, 4096], p=[0.4, 0.3, 0.2, 0.1])) for _ in range(2))
    if shape[0] * shape[1] > (1 << 23):
        shape = (shape[0], 512)
    lh = str(rng.choice(["poisson", "gaussian"]))
    pairs = int(rng.integers(2, 6))
    geo = bool(rng.random() < 0.3)
    dtype = torch.float64 if rng.random() < 0.7 else torch.float32
    if dtype == torch.float32:
        os.environ["NK_WIDE_FORWARD"] = "0"  # (the wide forward transform of fp32 models keeps the unbatched evaluation)
    else:
        os.environ.pop("NK_WIDE_FORWARD", None)
    t0 = time.time()
    try:
        a = T._iteration(shape, lh, pairs, {"NK_BATCH": "1"}, _Env(), geo=geo, dtype=dtype)
        b = T._iteration(shape, lh, pairs, {"NK_BATCH": "0"}, _Env(), geo=geo, dtype=dtype)
        T._same(a, b)
        res = "ok"
    except Exception as e:  # noqa: BLE001
        bad += 1
        res = "FAIL " + "".join(traceback.format_exception_only(type(e), e)).strip()[:300]
    print(f"{str(shape):14s} {lh:9s} pairs {pairs} geo {int(geo)} {str(dtype):14s} {time.time() - t0:6.1f}s  {res}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
