"""Developer probe for the GPU box: correctness of the raw C-ABI kernels vs torch/numpy + timings.
Usage: python tools/gpu_probe.py [quick|perf]"""
import ctypes
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from nifty_amd import _lib as L

lib = L.load()
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "quick"


def plan(shape, dtype, batch=1):
    p = ctypes.c_void_p()
    shp = (ctypes.c_int64 * len(shape))(*shape)
    L.check(lib.nk_plan_create(ctypes.byref(p), len(shape), shp, L.NK_F32 if dtype == torch.float32 else L.NK_F64, batch))
    ws = torch.empty(lib.nk_plan_workspace_bytes(p), dtype=torch.uint8, device=dev)
    return p, ws


def hartley(p, ws, x, conv=0):
    out = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.nk_hartley(p, x.data_ptr(), out.data_ptr(), 1.0, conv, ws.data_ptr(), st))
    return out


def ref_hartley(x, nd, conv=0):
    F = torch.fft.fftn(x.to(torch.float64), dim=tuple(range(-nd, 0)))
    return F.real + F.imag if conv == 0 else F.real - F.imag


ok = True
if mode in ("quick", "all"):
    for shape in [(2,), (8,), (512,), (4096,), (4, 4), (64, 64), (2048, 2048), (8, 4, 16), (64, 64, 64), (128, 256, 64)]:
        for dt in (torch.float64, torch.float32):
            x = torch.randn(shape, dtype=dt, device=dev)
            p, ws = plan(shape, dt)
            for conv in (0, 1):
                got = hartley(p, ws, x, conv)
                ref = ref_hartley(x, len(shape), conv)
                err = ((got.double() - ref).abs().max() / ref.abs().max()).item()
                tol = 1e-12 if dt == torch.float64 else 3e-5
                flag = "OK" if err < tol else "FAIL"
                ok &= err < tol
                print(f"hartley {shape} {dt} conv={conv} err={err:.2e} {flag}")
            # c2c
            xc = torch.randn(shape + (2,), dtype=dt, device=dev)
            out = torch.empty_like(xc)
            st = torch.cuda.current_stream().cuda_stream
            L.check(lib.nk_fftn(p, xc.data_ptr(), out.data_ptr(), 0, 1.0, ws.data_ptr(), st))
            refc = torch.fft.fftn(torch.view_as_complex(xc.double()), dim=tuple(range(len(shape))))
            err = ((torch.view_as_complex(out.double()) - refc).abs().max() / refc.abs().max()).item()
            ok &= err < (1e-12 if dt == torch.float64 else 3e-5)
            L.check(lib.nk_fftn(p, xc.data_ptr(), out.data_ptr(), 1, 1.0 / x.numel(), ws.data_ptr(), st))
            refi = torch.fft.ifftn(torch.view_as_complex(xc.double()), dim=tuple(range(len(shape))))
            err2 = ((torch.view_as_complex(out.double()) - refi).abs().max() / refi.abs().max()).item()
            ok &= err2 < (1e-12 if dt == torch.float64 else 3e-5)
            print(f"c2c {shape} {dt} fwd={err:.2e} inv={err2:.2e}")
            lib.nk_plan_destroy(p)
    # batch
    x = torch.randn((3, 16, 32), dtype=torch.float64, device=dev)
    p, ws = plan((16, 32), torch.float64, batch=3)
    err = (hartley(p, ws, x) - ref_hartley(x, 2)).abs().max().item()
    print("batch err", err)
    ok &= err < 1e-11
    # vec kernels
    n = 1_000_003
    for dt, code in ((torch.float64, 1), (torch.float32, 0)):
        a = torch.randn(n, dtype=dt, device=dev)
        b = torch.randn(n, dtype=dt, device=dev)
        res = torch.zeros(1, dtype=torch.float64, device=dev)
        L.check(lib.nk_vdot(n, a.data_ptr(), b.data_ptr(), code, res.data_ptr(), 0, 0))
        ref = (a.double() * b.double()).sum().item()
        print("vdot", dt, abs(res.item() - ref) / abs(ref))
        ok &= abs(res.item() - ref) < 1e-9 * abs(ref) + 1e-9
        o = torch.empty_like(a)
        L.check(lib.nk_axpby(n, 2.0, a.data_ptr(), -0.5, b.data_ptr(), o.data_ptr(), code, 0))
        ok &= (o - (2 * a - 0.5 * b)).abs().max().item() < 1e-5
        L.check(lib.nk_binary(3, n, a.data_ptr(), 0.0, b.data_ptr(), 0.0, o.data_ptr(), code, 0))
        ok &= torch.allclose(o, a / b, rtol=1e-5)
        fx = torch.empty_like(a); dfx = torch.empty_like(a)
        L.check(lib.nk_pointwise(4, 0.0, n, a.data_ptr(), fx.data_ptr(), dfx.data_ptr(), code, 0))
        ok &= torch.allclose(fx, 0.5 + 0.5 * torch.tanh(a), atol=1e-6)
        # cg kernels
        x0 = torch.randn(n, dtype=dt, device=dev); r0 = torch.randn(n, dtype=dt, device=dev)
        d0 = torch.randn(n, dtype=dt, device=dev); q0 = torch.randn(n, dtype=dt, device=dev)
        scal = torch.zeros(8, dtype=torch.float64, device=dev)
        scal[0] = 3.0
        xx, rr, dd = x0.clone(), r0.clone(), d0.clone()
        L.check(lib.nk_cg_curv(n, dd.data_ptr(), q0.data_ptr(), code, scal.data_ptr(), 0))
        L.check(lib.nk_cg_update(n, xx.data_ptr(), rr.data_ptr(), dd.data_ptr(), q0.data_ptr(), b.data_ptr(), code, scal.data_ptr(), 0))
        curv = (d0.double() * q0.double()).sum().item(); alpha = 3.0 / curv
        xr = x0.double() - alpha * d0.double(); rrf = r0.double() - alpha * q0.double()
        s = scal.cpu().numpy()
        print("cg", dt, abs(s[1] - curv) / abs(curv), abs(s[2] - (rrf * rrf).sum().item()) / s[2], abs(s[4] - (xr * b.double()).sum().item()))
        L.check(lib.nk_cg_direction(n, dd.data_ptr(), rr.data_ptr(), code, scal.data_ptr(), 0))
        beta = max(0.0, s[2] / 3.0)
        ok &= torch.allclose(dd.double(), beta * d0.double() + rrf, rtol=1e-4, atol=1e-4)
    # amplitude kernels vs oracle
    from oracle import nifty_oracle as orc
    for shape in [(128,), (64, 64), (32, 32, 32), (512, 512)]:
        cf = orc.CFModel(shape, None, orc.CFParams(offset_mean=2.0))
        rng = np.random.default_rng(5)
        x = cf.draw_latent(rng); x["xi"] = np.zeros(shape)
        v = cf.draw_latent(rng)
        st_o = cf.amplitude_state(x)
        nb = cf.geo.nb
        geo = np.concatenate([cf.rel, cf.sc, cf.mult, np.concatenate([cf.delta, [0, 0]])])
        hyp = np.array([*cf.ln["fluctuations"], *cf.ln["flexibility"], *cf.ln["asperity"], *cf.ln["zeromode"], *cf.slope_ms, cf.V])
        def pack(z):
            return np.concatenate([[z["asperity"], z["flexibility"], z["fluctuations"], z["loglogavgslope"], z["zeromode"]], z["spectrum"].ravel()]).astype(np.float64)
        t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
        g_d, h_d, lat_d, dlat_d = t(geo), t(hyp), t(pack(x)), t(pack(v))
        state = torch.zeros(8 * nb + 16, dtype=torch.float64, device=dev)
        amp = torch.empty(nb, dtype=torch.float64, device=dev); damp = torch.empty_like(amp)
        L.check(lib.nk_amp_forward(nb, g_d.data_ptr(), h_d.data_ptr(), lat_d.data_ptr(), state.data_ptr(), amp.data_ptr(), 0))
        e1 = np.max(np.abs(amp.cpu().numpy() - st_o["a"])) / np.max(np.abs(st_o["a"]))
        L.check(lib.nk_amp_jvp(nb, g_d.data_ptr(), h_d.data_ptr(), lat_d.data_ptr(), state.data_ptr(), dlat_d.data_ptr(), damp.data_ptr(), 0))
        da = cf.amplitude_jvp(st_o, v)
        e2 = np.max(np.abs(damp.cpu().numpy() - da)) / np.max(np.abs(da))
        abar = rng.normal(size=nb)
        latbar = torch.zeros(5 + 2 * (nb - 2), dtype=torch.float64, device=dev)
        L.check(lib.nk_amp_vjp(nb, g_d.data_ptr(), h_d.data_ptr(), lat_d.data_ptr(), state.data_ptr(), t(abar).data_ptr(), latbar.data_ptr(), 0))
        vj = cf.amplitude_vjp(st_o, abar)
        refb = pack(vj)
        e3 = np.max(np.abs(latbar.cpu().numpy() - refb)) / np.max(np.abs(refb))
        print(f"amp {shape} nb={nb} fwd={e1:.2e} jvp={e2:.2e} vjp={e3:.2e}")
        ok &= max(e1, e2, e3) < 1e-10
    print("ALL OK" if ok else "SOME FAILED")

if mode in ("perf", "all"):
    def timeit(fn, n=5):
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    for shape, dt in [((2048, 2048), torch.float64), ((4096, 4096), torch.float64), ((512, 512, 512), torch.float64),
                      ((1024, 1024, 1024), torch.float32), ((512, 512, 512), torch.float32)]:
        x = torch.randn(shape, dtype=dt, device=dev)
        p, ws = plan(shape, dt)
        out = torch.empty_like(x)
        st = torch.cuda.current_stream().cuda_stream
        ms = timeit(lambda: L.check(lib.nk_hartley(p, x.data_ptr(), out.data_ptr(), 1.0, 0, ws.data_ptr(), st)))
        nbytes = x.numel() * x.element_size() * 2 * len(shape)
        print(f"hartley {shape} {dt}: {ms:.3f} ms  -> {nbytes / ms / 1e6:.1f} GB/s (per-axis-pass model)")
        # copy ceiling
        msc = timeit(lambda: out.copy_(x))
        print(f"   copy: {msc:.3f} ms -> {2 * x.numel() * x.element_size() / msc / 1e6:.1f} GB/s")
        if len(shape) == 3 and dt == torch.float64 and shape[0] <= 512:
            mst = timeit(lambda: torch.fft.rfftn(x))
            print(f"   torch rfftn (rocFFT): {mst:.3f} ms")
        lib.nk_plan_destroy(p)
        del x, out, ws
        torch.cuda.empty_cache()
