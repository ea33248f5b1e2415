"""tests/test_engine_gpu.py::test_pair_final_pass_is_bit_identical (pair launches of the final pass against single launches:
same bits, single applications, KL metric, a Newton-CG step) over random 3-D grids with the sandwich pipeline, both field
types.  usage: python tools/gpu_pair_sweep.py [n] [seed]"""
import sys
import time
import traceback

import numpy as np
import torch

sys.path.insert(0, ".")
from tests.test_engine_gpu import test_pair_final_pass_is_bit_identical as check  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(n):
    while True:
        shape = tuple(int(rng.choice([64, 128, 256, 512, 1024], p=[0.35, 0.3, 0.2, 0.1, 0.05])) for _ in range(3))
        if np.prod(shape) <= (1 << 23) and shape[2] >= 128:  # (last axis 64: no sandwich pipeline, nothing to pair)
            break
    for dtype in (torch.float64, torch.float32):
        t0 = time.time()
        try:
            check(shape, dtype)
            res = "ok"
        except Exception as e:  # noqa: BLE001
            bad += 1
            tb = traceback.extract_tb(e.__traceback__)[-1]
            res = "FAIL " + "".join(traceback.format_exception_only(type(e), e)).strip()[:200] + f" at {tb.filename.split('/')[-1]}:{tb.lineno} {tb.line}"
        print(f"{str(shape):18s} {str(dtype):14s} {time.time() - t0:6.1f}s  {res}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
