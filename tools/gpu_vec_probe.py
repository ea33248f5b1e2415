"""Times the BLAS-1 kernels of the CG (nk_cg_update: 7 streams, nk_cg_direction: 3 streams, nk_vdot, nk_axpby) on one
latent-sized vector.  usage: python tools/gpu_vec_probe.py [n_elements] [f32|f64]"""
import sys
import torch
sys.path.insert(0, ".")
from nifty_amd import backend as B, _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
dt = torch.float32 if (len(sys.argv) < 3 or sys.argv[2] == "f32") else torch.float64
dev = torch.device("cuda:0")
x, r, d, q, b = (torch.randn(n, dtype=dt, device=dev) for _ in range(5))
scal = torch.ones(8, dtype=torch.float64, device=dev)
lib, st, code, bs = L.load(), B._stream(), B.dtype_code(x), x.element_size()


def timed(tag, streams, fn, reps=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{tag:14s} {ms:7.3f} ms  {streams * n * bs / ms / 1e6:7.1f} GB/s")


timed("cg_update", 7, lambda: L.check(lib.nk_cg_update(n, x.data_ptr(), r.data_ptr(), d.data_ptr(), q.data_ptr(), b.data_ptr(), code, scal.data_ptr(), 0, st)))
timed("cg_direction", 3, lambda: L.check(lib.nk_cg_direction(n, d.data_ptr(), r.data_ptr(), code, scal.data_ptr(), 0, st)))
timed("vdot", 2, lambda: B.vdot(x, r))
timed("axpby", 3, lambda: B.axpby(1.0, x, 0.5, r, out=q))

# the same reductions on the share of one of 8 ranks (8 of the 64 reduction units, nk_red_layout): what a rank of a sharded CG runs
unit = int(lib.nk_red_unit(n, code))
if unit:
    m = n // 8
    units = torch.zeros(4 * 64, dtype=torch.float64, device=dev)
    L.check(lib.nk_red_layout(unit, 8, 64, 8, 64, 0, units.data_ptr()))
    n_full, n = n, m
    xs, rs, ds, qs, bs_ = (t[:m] for t in (x, r, d, q, b))
    timed("shard cg_update", 7, lambda: L.check(lib.nk_cg_update(m, xs.data_ptr(), rs.data_ptr(), ds.data_ptr(), qs.data_ptr(), bs_.data_ptr(), code, scal.data_ptr(), 0, st)))
    timed("shard vdot", 2, lambda: L.check(lib.nk_vdot(m, xs.data_ptr(), rs.data_ptr(), code, scal.data_ptr(), 0, st)))
    lib.nk_red_layout(0, 0, 0, 0, 0, 0, 0)
