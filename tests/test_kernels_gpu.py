"""GPU tests of the raw C-ABI transforms against the golden vectors generated from the reference
(HartleyOperator / FFTOperator outputs) and against size-independent properties at large sizes."""
import numpy as np
import pytest
import torch

from tests import goldenlib as gl

pytestmark = pytest.mark.gpu

TAGS = ["16f64", "512f64", "64x64f64", "32x32x32f64", "64x64f32", "8x4x16f64", "2048f32"]


@pytest.mark.parametrize("tag", TAGS)
def test_hartley_and_fft_vs_reference_golden(tag):
    from nifty_amd import backend as B
    from nifty_amd import config

    z = gl.load("transforms")
    x = z[f"{tag}.x"]
    tol = 1e-5 if "f32" in tag else 1e-12
    xd = torch.from_numpy(x).cuda()
    try:
        for conv in ("non_canonical_hartley", "canonical_hartley"):
            config.update("hartley_convention", conv)
            got = B.hartley(xd).cpu().numpy()
            assert gl.relerr(got, z[f"{tag}.hartley.{conv}"]) < tol
    finally:
        config.update("hartley_convention", "non_canonical_hartley")
    xc = torch.from_numpy(z[f"{tag}.xc"]).cuda()
    # FFTOperator(harmonic->position): TIMES = N * ifftn, INVERSE_TIMES = fftn / N (harmonic_operators.py:77-94)
    got = B.fftn(xc, inverse=True, scale=1.0).cpu().numpy()
    assert gl.relerr(got, z[f"{tag}.fft"]) < tol
    got = B.fftn(xc, inverse=False, scale=1.0 / xc.numel()).cpu().numpy()
    assert gl.relerr(got, z[f"{tag}.ifft"]) < tol


@pytest.mark.parametrize("shape,dtype", [((1 << 13,), torch.float64), ((2048, 2048), torch.float64),
                                         ((4096, 512), torch.float32), ((256, 256, 256), torch.float64),
                                         ((512, 512, 512), torch.float32)])
def test_hartley_full_size_properties(shape, dtype):
    """Size-independent properties (reference test_fft_operator.py:38-115): H(H(x)) = N x (involution),
    zero mode = sum, linearity and Parseval."""
    from nifty_amd import backend as B

    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(shape, dtype=dtype, device="cuda", generator=g)
    y = torch.randn(shape, dtype=dtype, device="cuda", generator=g)
    n = x.numel()
    hx = B.hartley(x)
    tol = 1e-11 if dtype == torch.float64 else 2e-4
    back = B.hartley(hx, scale=1.0 / n)
    assert ((back - x).abs().max() / x.abs().max()).item() < tol
    assert abs(hx.reshape(-1)[0].item() - x.double().sum().item()) < tol * n ** 0.5 * 10
    lin = B.hartley(B.axpby(2.0, x, -3.0, y))
    ref = B.axpby(2.0, hx, -3.0, B.hartley(y))
    assert ((lin - ref).abs().max() / ref.abs().max()).item() < tol
    e1 = B.vdot(hx.reshape(-1), hx.reshape(-1)).item()
    e0 = B.vdot(x.reshape(-1), x.reshape(-1)).item()
    assert abs(e1 / n - e0) < tol * e0


def test_unsupported_shapes_raise():
    from nifty_amd import backend as B

    # lengths the planner rejects are served by the chirp-z composition (test_any_length_transforms) ...
    assert not B.plan_supported((22,), torch.float64, 1, "cuda:0")      # prime factor > 7
    assert not B.plan_supported((8, 15), torch.float64, 1, "cuda:0")    # odd last axis (real-to-complex packing)
    assert B.plan_supported((8, 14), torch.float64, 1, "cuda:0")
    # ... until the padded convolution length leaves the single-line limit
    with pytest.raises(NotImplementedError):
        B.hartley(torch.zeros(20011, device="cuda", dtype=torch.float64))
    with pytest.raises(RuntimeError):
        B.hartley(torch.zeros(16, dtype=torch.float64))


@pytest.mark.parametrize("shape", [(12,), (30,), (1000,), (6, 10), (15, 14), (120, 250), (100, 100), (9, 25, 28), (60, 50, 48),
                                   (192, 320), (96, 96, 96)])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_mixed_radix_transforms(shape, dtype):
    """Axis lengths with factors 2, 3, 5, 7 (ducc0 accepts any length; reference test_fft_operator.py:58-103 uses 10, 11, 12
    -- 11 goes through the chirp-z fallback, see test_any_length_transforms) against scipy.fft on the host."""
    import scipy.fft

    from nifty_amd import backend as B

    rng = np.random.default_rng(11)
    x = rng.normal(size=shape)
    xd = torch.from_numpy(x).to(dtype).cuda()
    F = scipy.fft.fftn(xd.cpu().numpy().astype(np.float64))
    tol = 1e-12 if dtype == torch.float64 else 3e-5
    got = B.hartley(xd).cpu().numpy()
    assert gl.relerr(got, F.real + F.imag) < tol
    xc = (rng.normal(size=shape) + 1j * rng.normal(size=shape))
    xcd = torch.from_numpy(xc).to(torch.complex128 if dtype == torch.float64 else torch.complex64).cuda()
    ref = scipy.fft.fftn(xcd.cpu().numpy().astype(np.complex128))
    got = B.fftn(xcd, inverse=False, scale=1.0).cpu().numpy()
    assert gl.relerr(got, ref) < tol
    got = B.fftn(xcd, inverse=True, scale=1.0 / xc.size).cpu().numpy()
    assert gl.relerr(got, scipy.fft.ifftn(xcd.cpu().numpy().astype(np.complex128))) < tol


@pytest.mark.parametrize("shape", [(11,), (15,), (1,), (10, 11), (11, 12), (13, 17, 6), (6, 15), (22, 64), (211,), (64, 1009),
                                   (3001,)])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_any_length_transforms(shape, dtype):
    """Lengths the native planner rejects (prime factors > 7, odd last axis): the array seam serves them through the
    chirp-z composition of native power-of-two transforms, like ducc0 takes every length (reference
    test_fft_operator.py:58-103 uses 10, 11, 12)."""
    import scipy.fft

    from nifty_amd import backend as B

    rng = np.random.default_rng(12)
    xd = torch.from_numpy(rng.normal(size=shape)).to(dtype).cuda()
    F = scipy.fft.fftn(xd.cpu().numpy().astype(np.float64))
    tol = 1e-12 if dtype == torch.float64 else 3e-5
    assert gl.relerr(B.hartley(xd).cpu().numpy(), F.real + F.imag) < tol
    assert gl.relerr(B.hartley(xd, scale=0.5).cpu().numpy(), 0.5 * (F.real + F.imag)) < tol
    xc = rng.normal(size=shape) + 1j * rng.normal(size=shape)
    xcd = torch.from_numpy(xc).to(torch.complex128 if dtype == torch.float64 else torch.complex64).cuda()
    xh = xcd.cpu().numpy().astype(np.complex128)
    assert gl.relerr(B.fftn(xcd, inverse=False, scale=1.0).cpu().numpy(), scipy.fft.fftn(xh)) < tol
    assert gl.relerr(B.fftn(xcd, inverse=True, scale=1.0 / xc.size).cpu().numpy(), scipy.fft.ifftn(xh)) < tol


@pytest.mark.parametrize("shape", [(64,), (32, 64), (64, 64, 64), (128, 64, 256), (256, 256, 256)])
def test_octant_expand_and_scatter(shape):
    """nk_octant_expand (full and compact) = PowerDistributor TIMES, nk_octant_scatter = its adjoint on the octant sums
    (distributors.py:106-127), on a real PowerSpace pindex."""
    import ctypes

    import nifty_amd as ift
    from nifty_amd import _lib as L
    from nifty_amd import backend as B

    hsp = ift.RGSpace(shape).get_default_codomain()
    ps = ift.PowerSpace(hsp)
    pidx = np.array(ps.pindex).astype(np.int32)
    nb = ps.shape[0]
    rng = np.random.default_rng(5)
    table = rng.normal(size=nb)
    pd, td = torch.from_numpy(pidx).cuda(), torch.from_numpy(table).cuda()
    shp = (ctypes.c_int64 * len(shape))(*shape)
    lib = L.load()
    full = torch.empty(shape, dtype=torch.float64, device="cuda")
    L.check(lib.nk_octant_expand(len(shape), shp, td.data_ptr(), pd.data_ptr(), full.data_ptr(), L.NK_F64, 0, B._stream()), "x")
    assert np.array_equal(full.cpu().numpy(), table[pidx])
    osl = tuple(slice(0, n // 2 + 1) for n in shape)
    oshape = tuple(n // 2 + 1 for n in shape)
    comp = torch.empty(oshape, dtype=torch.float64, device="cuda")
    L.check(lib.nk_octant_expand(len(shape), shp, td.data_ptr(), pd.data_ptr(), comp.data_ptr(), L.NK_F64, 1, B._stream()), "x")
    assert np.array_equal(comp.cpu().numpy(), table[pidx][osl])
    w8 = rng.normal(size=oshape)
    abar = torch.zeros(nb, dtype=torch.float64, device="cuda")
    merge = int(len(shape) == 3 and shape[0] == shape[1])
    w8d0 = torch.from_numpy(w8).cuda()
    L.check(lib.nk_octant_scatter(len(shape), shp, w8d0.data_ptr(), pd.data_ptr(), abar.data_ptr(), merge, B._stream()), "x")
    ref = np.bincount(pidx[osl].ravel(), weights=w8.ravel(), minlength=nb)
    assert np.max(np.abs(abar.cpu().numpy() - ref)) < 1e-12 * max(1.0, np.max(np.abs(ref)))
    # shell-binned variant for natural binning on equal-distance grids (overwrites abar)
    if len(set(hsp.distances)) == 1:
        k2 = np.nonzero(hsp._k2_flags())[0].astype(np.int32)
        assert len(k2) == nb
        scratch = torch.full((64 * (nb + 32),), np.nan, dtype=torch.float64, device="cuda")
        abar2 = torch.full((nb,), np.nan, dtype=torch.float64, device="cuda")
        w8d, k2d = torch.from_numpy(w8).cuda(), torch.from_numpy(k2).cuda()
        L.check(lib.nk_octant_scatter_k2(len(shape), shp, w8d.data_ptr(), pd.data_ptr(), k2d.data_ptr(), nb,
                                         scratch.data_ptr(), abar2.data_ptr(), None, B._stream()), "x")
        assert np.max(np.abs(abar2.cpu().numpy() - ref)) < 1e-12 * max(1.0, np.max(np.abs(ref)))
        # fixed-point accumulation (scale = max |w8|): same sums to 1e-13 of the largest contribution, and the same BITS on
        # every launch
        wmax = w8d.abs().max().reshape(1).clone()
        abar3 = torch.full((nb,), np.nan, dtype=torch.float64, device="cuda")
        L.check(lib.nk_octant_scatter_k2(len(shape), shp, w8d.data_ptr(), pd.data_ptr(), k2d.data_ptr(), nb,
                                         scratch.data_ptr(), abar3.data_ptr(), wmax.data_ptr(), B._stream()), "x")
        got3 = abar3.cpu().numpy()
        assert np.max(np.abs(got3 - ref)) < 1e-12 * max(1.0, float(wmax) * np.sqrt(w8.size / nb))
        for _ in range(3):
            abar3.fill_(np.nan)
            L.check(lib.nk_octant_scatter_k2(len(shape), shp, w8d.data_ptr(), pd.data_ptr(), k2d.data_ptr(), nb,
                                             scratch.data_ptr(), abar3.data_ptr(), wmax.data_ptr(), B._stream()), "x")
            assert np.array_equal(abar3.cpu().numpy(), got3)
        # all-zero input and a tiny scale
        w0 = torch.zeros_like(w8d)
        z = torch.zeros(1, dtype=torch.float64, device="cuda")
        L.check(lib.nk_octant_scatter_k2(len(shape), shp, w0.data_ptr(), pd.data_ptr(), k2d.data_ptr(), nb,
                                         scratch.data_ptr(), abar3.data_ptr(), z.data_ptr(), B._stream()), "x")
        assert not abar3.cpu().numpy().any()
        ws = w8d * 1e-200
        wm = ws.abs().max().reshape(1).clone()
        L.check(lib.nk_octant_scatter_k2(len(shape), shp, ws.data_ptr(), pd.data_ptr(), k2d.data_ptr(), nb,
                                         scratch.data_ptr(), abar3.data_ptr(), wm.data_ptr(), B._stream()), "x")
        assert np.max(np.abs(abar3.cpu().numpy() - 1e-200 * ref)) < 1e-12 * 1e-200 * max(1.0, float(wmax) * np.sqrt(w8.size / nb))


@pytest.mark.parametrize("n", [1, 7, 1024, 1025, 5000, 300001])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_cumsum(n, dtype):
    """nk_cumsum (prefix / suffix sums of _TwoLogIntegrations, correlated_fields.py:147-161) against numpy."""
    from nifty_amd import backend as B

    x = np.random.default_rng(n).normal(size=n)
    xd = torch.from_numpy(x).to(dtype).cuda()
    ref = np.cumsum(xd.cpu().numpy().astype(np.float64))
    tol = 1e-13 if dtype == torch.float64 else 2e-6
    scale = max(1.0, float(np.max(np.abs(ref))))
    assert np.max(np.abs(B.cumsum(xd).cpu().numpy() - ref)) < tol * scale
    refr = np.cumsum(xd.cpu().numpy().astype(np.float64)[::-1])[::-1]
    assert np.max(np.abs(B.cumsum(xd, reverse=True).cpu().numpy() - refr)) < tol * scale
    y = xd.clone()
    B.cumsum(y, out=y)  # in place
    assert np.max(np.abs(y.cpu().numpy() - ref)) < tol * scale


@pytest.mark.parametrize("shape,dtype", [((64, 128), torch.float64), ((2048, 2048), torch.float64), ((256, 128), torch.float32),
                                         ((64, 64, 128), torch.float64), ((128, 256, 256), torch.float32),
                                         ((256, 256, 256), torch.float64), ((64, 128, 4096), torch.float32),
                                         ((512, 512, 512), torch.float32)])
@pytest.mark.parametrize("field_mid", [False, True])
def test_sandwich_equals_two_transforms(shape, dtype, field_mid):
    """nk_hartley_sandwich (five passes, no position-space intermediate; nk_fft3.h) against the same H D H computed
    with two nk_hartley_fused calls, plain prologue / affine epilogue, both conventions."""
    from nifty_amd import _lib as L
    from nifty_amd import backend as B
    from nifty_amd import config

    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(shape, dtype=dtype, device="cuda", generator=g)
    m = torch.randn(shape, dtype=dtype, device="cuda", generator=g) if field_mid else None
    plan = B.get_plan(shape, dtype, 1, x.device)
    assert B.plan_sandwich(plan)
    tol = 1e-11 if dtype == torch.float64 else 3e-4
    n = x.numel()
    try:
        for conv in ("non_canonical_hartley", "canonical_hartley"):
            config.update("hartley_convention", conv)
            tmp, ref, out = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
            f = L.Fuse()
            f.pro, f.in_, f.epi, f.out, f.scale = L.PRO_PLAIN, x.data_ptr(), L.EPI_MUL, tmp.data_ptr(), 0.25
            f.mul, f.mul_scalar = B.ptr(m), 1.5
            B.hartley_fused(plan, f)
            f = L.Fuse()
            f.pro, f.in_, f.epi, f.out, f.scale, f.offset = L.PRO_PLAIN, tmp.data_ptr(), L.EPI_AFFINE, ref.data_ptr(), 2.0 / n, 0.5
            B.hartley_fused(plan, f)
            f = L.Fuse()
            f.pro, f.in_, f.epi, f.out, f.scale, f.offset = L.PRO_PLAIN, x.data_ptr(), L.EPI_AFFINE, out.data_ptr(), 2.0 / n, 0.5
            f.mul, f.mul_scalar = B.ptr(m), 1.5
            B.hartley_sandwich(plan, f, 0.25)
            err = ((out - ref).abs().max() / ref.abs().max()).item()
            assert err < tol, (conv, err)
    finally:
        config.update("hartley_convention", "non_canonical_hartley")


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_cg_update_dr(dtype):
    """nk_cg_update_dr: the fused CG update that returns d.r (old residual) instead of x.r / x.b."""
    from nifty_amd import _lib as L
    from nifty_amd import backend as B

    n = 1 << 20
    g = torch.Generator(device="cuda").manual_seed(5)
    x, r, d, q = (torch.randn(n + 3, dtype=dtype, device="cuda", generator=g) for _ in range(4))
    scal = torch.tensor([3.0, 2.0, 0, 0, 0, 0, 0, 0], dtype=torch.float64, device="cuda")
    alpha = 1.5
    xr, rr = (x.double() - alpha * d.double()).to(dtype), (r.double() - alpha * q.double()).to(dtype)
    dr = float((d.double() * r.double()).sum())
    L.check(L.load().nk_cg_update_dr(n + 3, x.data_ptr(), r.data_ptr(), d.data_ptr(), q.data_ptr(), B.dtype_code(x),
                                     scal.data_ptr(), 0, B._stream()))
    tol = 4e-16 if dtype == torch.float64 else 1e-7  # the kernel contracts x - alpha d into one fma
    assert ((x - xr).abs().max() <= tol * 8).item() and ((r - rr).abs().max() <= tol * 8).item()
    s = scal.cpu().numpy()
    assert abs(s[2] - float((r.double() ** 2).sum())) < 1e-12 * s[2]
    assert abs(s[3] - dr) < 1e-10 * abs((d.double() * r.double()).abs().sum().item())


@pytest.mark.parametrize("shape,dtype", [((1024, 1024, 1024), torch.float32), ((512, 512, 512), torch.float64)])
def test_sandwich_full_size_properties(shape, dtype):
    """Size-independent properties of the five-pass H D H at BASELINE's full sizes: with D = 1/N it is the identity
    (H o H = N), and with a positive diagonal it is symmetric, <y, S x> = <x, S y>."""
    from nifty_amd import _lib as L
    from nifty_amd import backend as B

    g = torch.Generator(device="cuda").manual_seed(17)
    x = torch.randn(shape, dtype=dtype, device="cuda", generator=g)
    plan = B.get_plan(shape, dtype, 1, x.device)
    assert B.plan_sandwich(plan)
    n = x.numel()
    tol = 1e-11 if dtype == torch.float64 else 3e-4

    def sandwich(inp, mid):
        out = torch.empty_like(inp)
        f = L.Fuse()
        f.pro, f.in_, f.epi, f.out, f.scale, f.mul_scalar, f.mul = L.PRO_PLAIN, inp.data_ptr(), L.EPI_AFFINE, out.data_ptr(), 1.0 / n, 1.0, B.ptr(mid)
        B.hartley_sandwich(plan, f, 1.0)
        return out

    back = sandwich(x, None)
    assert ((back - x).abs().max() / x.abs().max()).item() < tol
    del back
    y = torch.randn(shape, dtype=dtype, device="cuda", generator=g)
    mid = torch.rand(shape, dtype=dtype, device="cuda", generator=g) + 0.5
    sx = sandwich(x, mid)
    a = B.vdot(y.reshape(-1), sx.reshape(-1)).item()
    del sx
    sy = sandwich(y, mid)
    b = B.vdot(x.reshape(-1), sy.reshape(-1)).item()
    assert abs(a - b) < tol * (abs(a) + abs(b) + n ** 0.5)


@pytest.mark.parametrize("n,nb", [(1, 1), (1000, 37), (1025 * 1025, 313847), (300000, 5)])
def test_segment_sum_equals_bincount(n, nb):
    """nk_segment_sum over the bin-sorted permutation == np.bincount(pindex, weights) (distributors.py:106-127 adjoint)."""
    from nifty_amd import _lib as L
    from nifty_amd import backend as B

    rng = np.random.default_rng(n + nb)
    pidx = rng.integers(0, nb, size=n).astype(np.int32)
    if nb > 2:
        pidx[pidx == 1] = 0  # an empty bin
    w = rng.normal(size=n)
    ref = np.bincount(pidx, weights=w, minlength=nb)
    perm = np.argsort(pidx, kind="stable").astype(np.int32)
    rowptr = np.zeros(nb + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(np.bincount(pidx, minlength=nb))
    dev = torch.device("cuda:0")
    wd, pd, rd = torch.from_numpy(w).to(dev), torch.from_numpy(perm).to(dev), torch.from_numpy(rowptr).to(dev)
    out = torch.full((nb,), 7.0, dtype=torch.float64, device=dev)
    L.check(L.load().nk_segment_sum(nb, rd.data_ptr(), pd.data_ptr(), wd.data_ptr(), out.data_ptr(), 0, B._stream()))
    got = out.cpu().numpy()
    assert np.allclose(got, ref, rtol=1e-13, atol=1e-13 * np.abs(w).sum() / max(nb, 1))
    L.check(L.load().nk_segment_sum(nb, rd.data_ptr(), pd.data_ptr(), wd.data_ptr(), out.data_ptr(), 1, B._stream()))
    assert np.allclose(out.cpu().numpy(), 2 * got, rtol=1e-15, atol=0)
    # same bits on a second launch (fixed summation order)
    out2 = torch.empty_like(out)
    L.check(L.load().nk_segment_sum(nb, rd.data_ptr(), pd.data_ptr(), wd.data_ptr(), out2.data_ptr(), 0, B._stream()))
    assert np.array_equal(out2.cpu().numpy(), got)


def test_deterministic_mode_is_bit_reproducible():
    """Two runs (child processes) of an MGVI step on a 3-D grid give identical bits -- every sum of the hot path is built in a
    fixed order or in fixed point."""
    import os
    import subprocess
    import sys

    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from nifty_amd import random
from nifty_amd.engine import FusedModel, mgvi_iteration
from nifty_amd.minimization import AbsDeltaEnergyController, NewtonCG
model = FusedModel((64, 128, 128), offset_mean=2.0, likelihood="gaussian", icov=100.0, dtype=torch.float32, device="cuda:0")
random.push_sseq_from_seed(3)
truth = model.draw_prior()
model.set_data(model.signal(truth), 100.0)
mean = 0.1 * model.draw_prior()
ic = lambda: AbsDeltaEnergyController(0.05, iteration_limit=5)
mini = NewtonCG(AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=2), max_cg_iterations=5)
mean, kl = mgvi_iteration(model, mean, 2, ic, mini, mirror_samples=True)
print("RESULT", repr(float(kl.value)), repr(float(mean.xi.double().sum())), repr(float(mean.small.sum())))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(det):
        env = dict(os.environ, NK_SCATTER_FP_ATOMICS="0" if det else "1")
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")][-1]
        return [float(v) for v in line.split()[1:]]

    a, b = run(True), run(True)  # (the floating-point-atomic kernel is covered by test_octant_expand_and_scatter above)
    assert a == b


def test_fence_free_reductions_equal_the_fenced_build():
    """The ticket-ordered reductions publish their block partials with agent-scope atomics instead of device-scope fences
    (nk_util.h; ADVICE r5: that leans on where this chip performs them).  `make fence` builds the same library with the
    textbook fence protocol (-DNK_RED_FENCE=1); both run a stress mix -- dot products, CG updates and amplitude sums over
    many launches, lengths (1 ... 2^24 + 5: one workgroup ... the full unit grid) and both dtypes, interleaved on one
    stream so that a stale partial or ticket of the previous launch would show -- and must give the same bits."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fence = os.path.join(root, "build", "libniftyk_fence.so")
    srcs = [os.path.join(root, "nifty_amd", "csrc", f) for f in ("nk_vec.hip", "nk_amp.hip", "nk_util.h", "libniftyk.so")]
    if not os.path.exists(fence) or os.path.getmtime(fence) < max(os.path.getmtime(f) for f in srcs):
        made = subprocess.run(["make", "-C", os.path.join(root, "nifty_amd", "csrc"), "fence", "ARCH=gfx950"], capture_output=True,
                              text=True, timeout=900)
        assert made.returncode == 0, made.stderr[-2000:]
    code = r"""
import sys, hashlib, numpy as np, torch
sys.path.insert(0, %r)
from nifty_amd import backend as B
from nifty_amd.engine import FusedModel
h = hashlib.sha256()
gen = torch.Generator(device="cuda").manual_seed(11)
for rep in range(6):
    for n in (1, 100, 4097, 65539, 1 << 20, (1 << 22) + 5, 1 << 24, (1 << 24) + 5):
        for dt in (torch.float64, torch.float32):
            a = torch.randn(n, generator=gen, device="cuda", dtype=dt)
            b = torch.randn(n, generator=gen, device="cuda", dtype=dt)
            h.update(B.vdot(a, b).cpu().numpy().tobytes())
            h.update(B.vdot(b, b).cpu().numpy().tobytes())
model = FusedModel((256, 320), offset_mean=2.0, likelihood="gaussian", icov=100.0, dtype=torch.float64, device="cuda:0")
for rep in range(20):
    x = 0.1 * model.draw_prior(gen)
    model.set_data(model.signal(x), 100.0)
    lp = model.linearize(x)
    q = model.metric(lp, model.draw_prior(gen))
    h.update(lp.value.cpu().numpy().tobytes()); h.update(lp.grad.small.cpu().numpy().tobytes()); h.update(q.small.cpu().numpy().tobytes())
print("RESULT", h.hexdigest())
""" % root

    def run(lib=None):
        env = dict(os.environ, **({"NK_LIB_PATH": lib} if lib else {}))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        return [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")][-1]

    assert run() == run(fence)


def test_launch_variants_of_the_metric_agree():
    """One metric application of a 3-D model through every selectable variant of its kernels -- QUAD / row-tile launches of the
    contiguous first pass (NK_CONTIG_QUAD 1 / 0 / 2: library statics, hence child processes), `da` expanded in sorted or
    natural line order or gathered in the prologue (class 7) -- gives the same bits for the launch shapes and the line
    orders, and agrees to rounding where the arithmetic differs (the gather multiplies by the fp32-rounded table entry like
    the expansion does: same bits too)."""
    import os
    import subprocess
    import sys

    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from nifty_amd import random
from nifty_amd.engine import FusedModel, LatentVec
model = FusedModel((64, 128, 128), offset_mean=2.0, likelihood="gaussian", icov=100.0, dtype=torch.float32, device="cuda:0")
random.push_sseq_from_seed(5)
truth = model.draw_prior()
model.set_data(model.signal(truth), 100.0)
x, d = 0.1 * model.draw_prior(), model.draw_prior()
lp = model.linearize(x)
q = model.metric(lp, d)
print("RESULT", repr(float(lp.value.item())), repr(float(q.xi.double().sum())), repr(float((q.xi.double() ** 2).sum())),
      repr(float(q.small.sum())))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(**env):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True,
                             timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")][-1]
        return [float(v) for v in line.split()[1:]]

    base = run()
    for env in (dict(NK_CONTIG_QUAD="0"), dict(NK_CONTIG_QUAD="2"), dict(NK_EXPAND_ORDER="0"), dict(NK_EXPAND_K2="0"),
                dict(NK_DA_GATHER="1")):
        assert run(**env) == base, env


@pytest.mark.parametrize("lanes", [1, 4, 16, 64])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_csr_rowsum_against_scipy_and_bitwise_repeatable(lanes, dtype):
    """nk_csr_rowsum (weighted = LOSResponse products, unweighted = bin sums of a static index map) against scipy.sparse /
    np.bincount, every lanes-per-row variant on rows of 0 .. 300 entries; the same bits on every launch."""
    from scipy.sparse import random as sprandom

    from nifty_amd import _lib as L
    from nifty_amd import backend as B

    rng = np.random.default_rng(lanes)
    m = sprandom(3000, 5000, density=0.01, format="csr", random_state=7, dtype=np.float32)
    m = m[np.r_[0:1500, 1500:3000:2]]  # plus a run of rows in a different order
    x = rng.normal(size=5000).astype(np.float32 if dtype == torch.float32 else np.float64)
    dev = torch.device("cuda:0")
    rp, col, wg = (torch.from_numpy(a).to(dev) for a in (m.indptr.astype(np.int64), m.indices.astype(np.int32), m.data))
    xd = torch.from_numpy(x).to(dev)
    y = torch.empty(m.shape[0], dtype=dtype, device=dev)
    lib = L.load()
    L.check(lib.nk_csr_rowsum(m.shape[0], rp.data_ptr(), col.data_ptr(), wg.data_ptr(), xd.data_ptr(), y.data_ptr(),
                              B.dtype_code(xd), lanes, B._stream()))
    ref = m.astype(np.float64) @ x.astype(np.float64)
    tol = 1e-13 if dtype == torch.float64 else 2e-6
    assert np.max(np.abs(y.cpu().numpy() - ref)) < tol * max(1.0, np.max(np.abs(ref)))
    y2 = torch.empty_like(y)
    L.check(lib.nk_csr_rowsum(m.shape[0], rp.data_ptr(), col.data_ptr(), wg.data_ptr(), xd.data_ptr(), y2.data_ptr(),
                              B.dtype_code(xd), lanes, B._stream()))
    assert torch.equal(y, y2)
    # unweighted: bin sums through bin_plan / bin_sum == np.bincount
    nb = 777
    pidx = rng.integers(0, nb, size=200000).astype(np.int32)
    pidx[pidx == 5] = 6  # an empty bin
    w = rng.normal(size=pidx.size).astype(x.dtype)
    plan = B.bin_plan(torch.from_numpy(pidx).to(dev), nb)
    plan = (plan[0], plan[1], lanes)
    got = B.bin_sum(torch.from_numpy(w).to(dev), plan)
    refb = np.bincount(pidx, weights=w.astype(np.float64), minlength=nb)
    assert np.max(np.abs(got.cpu().numpy() - refb)) < (1e-12 if dtype == torch.float64 else 1e-5) * np.max(np.abs(refb))
    assert torch.equal(got, B.bin_sum(torch.from_numpy(w).to(dev), plan))
    with pytest.raises(ValueError):
        L.check(lib.nk_csr_rowsum(1, rp.data_ptr(), col.data_ptr(), 0, xd.data_ptr(), y.data_ptr(), B.dtype_code(xd), 3,
                                  B._stream()))


def test_fixed_point_scatter_answers_non_finite_input_with_nan():
    """a NaN / Inf among the octant sums must not come out of the fixed-point shell scatter as a finite number.
    The final pass joins max |w8| with a NaN-propagating rule; the scatter then writes NaN.  Through the engine: a NaN
    in the data gives a NaN (-> inf) energy and a NaN spectrum gradient, like the floating-point path."""
    import ctypes

    import nifty_amd as ift
    from nifty_amd import _lib as L
    from nifty_amd import backend as B
    from nifty_amd import random
    from nifty_amd.engine import FusedModel

    shape = (64, 64, 64)
    hsp = ift.RGSpace(shape).get_default_codomain()
    ps = ift.PowerSpace(hsp)
    pidx = torch.from_numpy(np.array(ps.pindex).astype(np.int32)).cuda()
    nb = ps.shape[0]
    k2d = torch.from_numpy(np.nonzero(hsp._k2_flags())[0].astype(np.int32)).cuda()
    oshape = tuple(n // 2 + 1 for n in shape)
    shp = (ctypes.c_int64 * 3)(*shape)
    scratch = torch.empty(64 * (nb + 32), dtype=torch.float64, device="cuda")
    w8 = torch.randn(oshape, dtype=torch.float64, device="cuda")
    for bad in (float("nan"), float("inf")):
        wmax = torch.tensor([bad], dtype=torch.float64, device="cuda")
        abar = torch.zeros(nb, dtype=torch.float64, device="cuda")
        L.check(L.load().nk_octant_scatter_k2(3, shp, w8.data_ptr(), pidx.data_ptr(), k2d.data_ptr(), nb, scratch.data_ptr(),
                                              abar.data_ptr(), wmax.data_ptr(), B._stream()))
        assert bool(torch.isnan(abar).all())
    # end to end: a NaN in the data reaches the spectrum gradient
    model = FusedModel(shape, offset_mean=2.0, likelihood="gaussian", icov=100.0, dtype=torch.float64, device="cuda:0")
    assert model.scatter_fixed_point
    random.push_sseq_from_seed(9)
    try:
        x = model.draw_prior() * 0.1
        data = model.signal(x)
    finally:
        random.pop_sseq()
    data[3, 4, 5] = float("nan")
    model.set_data(data, 100.0)
    lp = model.linearize(x)
    assert not np.isfinite(float(lp.value.item()))
    assert bool(torch.isnan(lp.grad.small[5:]).any())


@pytest.mark.parametrize("shape", [(64, 128, 256), (256, 64, 64)])
def test_anisotropic_grid_scatter_eligibility_is_counted_exactly(shape):
    """the overflow bound of the fixed-point scatter (< 2^18 points per (shell, split) workgroup) is counted
    from the bin index of the grid at hand, not from a cube heuristic; the VJP bin sums of a non-cubic grid with equal
    physical box lengths agree with the host bincount."""
    from nifty_amd import random
    from nifty_amd.engine import FusedModel

    model = FusedModel(shape, offset_mean=1.0, likelihood="gaussian", icov=10.0, dtype=torch.float64, device="cuda:0")
    if model.bin_k2 is not None:
        a = np.arange(shape[0] // 2 + 1)
        p8 = model.pidx8.cpu().numpy().reshape(shape[0] // 2 + 1, -1)
        key = (p8 // 4096) * 16 + (a % 16)[:, None]
        assert model.scatter_busiest == int(np.bincount(key.ravel()).max())
        assert model.scatter_fixed_point == (model.scatter_busiest < (1 << 18))
    random.push_sseq_from_seed(2)
    try:
        x = model.draw_prior() * 0.1
        model.set_data(model.signal(model.draw_prior()), 10.0)
    finally:
        random.pop_sseq()
    lp = model.linearize(x)
    assert np.isfinite(float(lp.value.item()))
    # abar of the last VJP against the host scatter of the octant sums it was reduced from
    osl = tuple(slice(0, n // 2 + 1) for n in shape)
    pid8 = model.pidx.view(shape)[osl].cpu().numpy().ravel()
    ref = np.bincount(pid8, weights=model.w8.cpu().numpy().ravel(), minlength=model.nb)
    got = model.abar.cpu().numpy()
    assert np.max(np.abs(got - ref)) < 1e-11 * max(1.0, np.max(np.abs(ref)))


def test_reductions_on_concurrent_streams_do_not_share_scratch():
    """the ticket-ordered reductions keep their block partials per (device, stream).  Dots launched alternately
    on two streams (so that launches of one overlap launches of the other) equal the serial results bit for bit."""
    from nifty_amd import _lib as L

    n = 1 << 24
    g = torch.Generator(device="cuda").manual_seed(1)
    a = [torch.randn(n, dtype=torch.float32, device="cuda", generator=g) for _ in range(4)]
    lib = L.load()
    serial = torch.zeros(4, dtype=torch.float64, device="cuda")
    st0 = torch.cuda.current_stream().cuda_stream
    for i in range(4):
        L.check(lib.nk_vdot(n, a[i].data_ptr(), a[(i + 1) % 4].data_ptr(), L.NK_F32, serial[i:i + 1].data_ptr(), 0, st0))
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(20):
        out = torch.zeros(4, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        for i in range(4):
            st = (s1 if i % 2 == 0 else s2).cuda_stream
            L.check(lib.nk_vdot(n, a[i].data_ptr(), a[(i + 1) % 4].data_ptr(), L.NK_F32, out[i:i + 1].data_ptr(), 0, st))
        torch.cuda.synchronize()
        assert torch.equal(out, serial), rep


@pytest.mark.parametrize("shape,dtype", [((64, 64, 128), torch.float64), ((128, 64, 256), torch.float32)])
def test_staged_sandwich_is_bit_identical(shape, dtype):
    """nk_fuse.pipe_chunks: the first and the final pass of the sandwich in C/2 stages (slab pipelining against the
    exchange of the sharded CG) write the same bits as the one-launch passes -- with and without the event hand-over."""
    import ctypes

    from nifty_amd import _lib as L
    from nifty_amd import random
    from nifty_amd.engine import FusedModel, LatentVec

    model = FusedModel(shape, offset_mean=2.0, likelihood="gaussian", icov=100.0, dtype=dtype, device="cuda:0")
    assert model.sandwich
    random.push_sseq_from_seed(4)
    try:
        x = model.draw_prior() * 0.1
        model.set_data(model.signal(model.draw_prior()), 100.0)
        d = model.draw_prior()
    finally:
        random.pop_sseq()
    lp = model.linearize(x)

    def run(pipe, accumulate_into=None):
        out = LatentVec(torch.zeros_like(d.xi), None) if accumulate_into is None else accumulate_into
        dot = torch.zeros(1, dtype=torch.float64, device="cuda")
        model.lh_metric_accumulate(lp, d, out, 0.5, accumulate_into is None, identity=1.0, dot_out=dot, pipe=pipe)
        return out.xi.clone(), out.small.clone(), dot.clone(), model.abar.clone()

    ref = run(None)
    for C in (2, 4, 8, 16):
        assert L.load().nk_plan_pipe_ok(model.plan.handle, C) == 1
        got = run((C, None, None))
        for a, b in zip(got, ref):
            assert torch.equal(a, b), C
        # events: `wait` recorded on a side stream before the call, `record` consumed by a side stream after it
        ns = C // 2
        side = torch.cuda.Stream()
        ev_in, ev_out = [torch.cuda.Event() for _ in range(ns)], [torch.cuda.Event() for _ in range(ns)]
        for ev in ev_in + ev_out:
            ev.record()
        with torch.cuda.stream(side):
            for ev in ev_in:
                torch.cuda._sleep(200000)
                ev.record(side)
        h_in = (ctypes.c_void_p * ns)(*[ev.cuda_event for ev in ev_in])
        h_out = (ctypes.c_void_p * ns)(*[ev.cuda_event for ev in ev_out])
        got = run((C, h_in, h_out))
        for ev in ev_out:
            side.wait_event(ev)
        torch.cuda.synchronize()
        for a, b in zip(got, ref):
            assert torch.equal(a, b), C
    assert L.load().nk_plan_pipe_ok(model.plan.handle, 3) == 0 and L.load().nk_plan_pipe_ok(model.plan.handle, 6) == 0
    with pytest.raises(NotImplementedError):
        run((6, None, None))  # 6 does not divide the first axis


@pytest.mark.parametrize("shape", [(60, 50, 48), (100, 30), (30,)])
def test_mixed_radix_grids_are_reproducible_and_sum_their_bins_in_order(shape):
    """Plans without the octant pipeline (generic LDS kernels): the VJP epilogue deposits xi . t per grid point
    (nk_fuse.wfull) and the bins are summed in a fixed order, the energy goes through per-wavefront slots -- the spectrum
    gradient equals the host bincount of the deposited products and repeated evaluations give identical bits (until round 3
    these grids used fp64 atomics)."""
    from nifty_amd import random
    from nifty_amd.engine import FusedModel

    model = FusedModel(shape, offset_mean=1.0, likelihood="poisson", nonlin="exp", dtype=torch.float64, device="cuda:0")
    assert not model.octant_vjp and model.full_plan is not None
    random.push_sseq_from_seed(12)
    try:
        x = model.draw_prior() * 0.2
        model.set_data(torch.poisson(model.signal(model.draw_prior() * 0.3)).to(torch.int64))
        d = model.draw_prior()
    finally:
        random.pop_sseq()
    lp = model.linearize(x)
    ref = np.bincount(model.pidx.cpu().numpy().ravel(), weights=model.wfull.cpu().numpy(), minlength=model.nb)
    assert np.max(np.abs(model.abar.cpu().numpy() - ref)) < 1e-12 * max(1.0, np.max(np.abs(ref)))
    first = (float(lp.value.item()), lp.grad.xi.clone(), lp.grad.small.clone(), model.metric(lp, d))
    for _ in range(3):
        lp2 = model.linearize(x)
        q2 = model.metric(lp2, d)
        assert float(lp2.value.item()) == first[0]
        assert torch.equal(lp2.grad.xi, first[1]) and torch.equal(lp2.grad.small, first[2])
        assert torch.equal(q2.xi, first[3].xi) and torch.equal(q2.small, first[3].small)


@pytest.mark.parametrize("shape,dtype", [((64, 64, 64), torch.float64), ((128, 64, 256), torch.float32), ((256, 64, 128), torch.float64)])
def test_octant_expand_from_k2_table(shape, dtype):
    """nk_octant_expand_k2 (PowerDistributor TIMES on natural binning, distributors.py:114-119, without the index stream) in
    natural line order and in the order sorted by a^2 + b^2 against the gather through the int32 bin index: identical bits."""
    from nifty_amd.engine import FusedModel

    dist = tuple(1.0 / n for n in shape)  # equal harmonic distances on every axis: natural bins are functions of k^2
    model = FusedModel(shape, distances=dist, offset_mean=1.0, likelihood="gaussian", icov=1.0, dtype=dtype, device="cuda:0")
    assert model.k2_dense is not None and model.k2_line_order is not None
    order = model.k2_line_order.to(torch.int64)
    lines = (shape[0] // 2 + 1) * (shape[1] // 2 + 1)
    assert torch.equal(torch.sort(order).values, torch.arange(lines, device=order.device))  # a permutation of the lines
    table = torch.randn(model.nb, dtype=torch.float64, device="cuda:0")
    sorted_order = model._amp_field(table).clone()
    model.k2_line_order = None
    natural = model._amp_field(table).clone()
    dense, model.k2_dense = model.k2_dense, None
    gathered = model._amp_field(table).clone()
    model.k2_dense = dense
    assert torch.equal(natural, gathered)
    assert torch.equal(sorted_order, gathered)
