"""Transforms of lengths the native planner rejects (prime factors > 7): the one-launch chirp-z kernel (nk_bluestein_rows)
against the composition of three power-of-two c2c transforms + three element-wise launches per axis (NK_BLUESTEIN=0), HIP
events around B.hartley / B.fftn."""
import os
import sys

import torch

sys.path.insert(0, ".")
from nifty_amd import backend as B


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for shape in ((4096, 19), (4096, 73), (4096, 211), (2048, 1009), (1009, 1009), (211, 211, 211), (73, 73, 73), (500, 1009)):
    for dt in (torch.float64, torch.float32):
        x = torch.randn(shape, dtype=dt, device="cuda")
        res = {}
        for flag in ("1", "0"):
            os.environ["NK_BLUESTEIN"] = flag
            nd = 1 if shape[0] in (4096, 2048, 500) else len(shape)
            res[flag] = timed(lambda: B.hartley(x, ndim=nd))
        os.environ["NK_BLUESTEIN"] = "1"
        a = B.hartley(x, ndim=nd)
        os.environ["NK_BLUESTEIN"] = "0"
        b = B.hartley(x, ndim=nd)
        print(f"{str(shape):18s} {str(dt)[6:]:8s} axes {nd}  one launch {res['1']:9.1f} us   composed {res['0']:9.1f} us   x{res['0'] / res['1']:.1f}"
              f"   max diff {float((a - b).abs().max() / b.abs().max()):.1e}")
