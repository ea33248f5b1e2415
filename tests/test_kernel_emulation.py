"""Host emulation of the transform kernels (tests/emu): the SAME phase functions the GPU runs, executed
thread-by-thread on the CPU, against scipy.fft.  Checks index math, digit reversal, twiddles, the packed
half-spectrum layout and the fused prologue/epilogue without a GPU."""
import ctypes
import os

import numpy as np
import pytest
import scipy.fft

from nifty_amd._lib import Fuse

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu", "libnk_emu.so")

pytestmark = pytest.mark.skipif(not os.path.exists(EMU), reason="emulation library not built (run __graft_entry__.build())")


def lib():
    return ctypes.CDLL(EMU)


def run(f, shape, dtype, conv=0, batch=1, fn="emu_hartley_fused"):
    """fn: emu_hartley_fused = generic kernels; emu2_/emu3_ = register-resident fast path (1-D / strided-first).  Long first
    axes of 2-D grids run the two-level pass in every split the library has (NK_TWO_LEVEL=2: 64 x 64 and 64 x 32, both
    dtypes; the emulation reads the switch per call)."""
    shp = (ctypes.c_int64 * len(shape))(*shape)
    old = os.environ.get("NK_TWO_LEVEL")
    os.environ["NK_TWO_LEVEL"] = "2"
    try:
        rc = getattr(lib(), fn)(len(shape), shp, 0 if dtype == np.float32 else 1, batch, ctypes.byref(f), conv)
    finally:
        if old is None:
            del os.environ["NK_TWO_LEVEL"]
        else:
            os.environ["NK_TWO_LEVEL"] = old
    assert rc == 0, rc


def fast_fn(shape):
    return "emu2_hartley_fused" if len(shape) == 1 else "emu3_hartley_fused"


@pytest.mark.parametrize("shape", [(2,), (8,), (64,), (1024,), (2, 2), (4, 16), (32, 8), (64, 64), (2, 2, 2), (8, 4, 16),
                                   (16, 16, 16), (2, 32, 4),
                                   # mixed radix 2/3/5/7 (generic kernels; reference test_fft_operator.py:58-103 sizes)
                                   (6,), (10,), (12,), (30,), (210,), (98,), (500,), (3, 4), (6, 10), (15, 14), (50, 18),
                                   (5, 7, 12), (10, 6, 12), (9, 25, 28)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_hartley_emulation(shape, dtype):
    rng = np.random.default_rng(0)
    x = rng.normal(size=shape).astype(dtype)
    F = scipy.fft.fftn(x.astype(np.float64))
    for conv, ref in ((0, F.real + F.imag), (1, F.real - F.imag)):
        out = np.empty_like(x)
        f = Fuse()
        f.in_, f.out, f.scale = x.ctypes.data, out.ctypes.data, 1.0
        run(f, shape, dtype, conv)
        err = np.max(np.abs(out - ref)) / np.max(np.abs(ref))
        assert err < (1e-12 if dtype == np.float64 else 2e-5)


@pytest.mark.parametrize("shape", [(128,), (64, 128), (128, 64), (64, 64, 64), (64, 128, 64),
                                   (2048, 64), (4096, 64)])  # long first axes of 2-D grids: the two-level pass (nk_tl_split)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fast_path_emulation(shape, dtype):
    rng = np.random.default_rng(3)
    x = rng.normal(size=(2,) + shape).astype(dtype)  # batch of 2
    F = scipy.fft.fftn(x.astype(np.float64), axes=tuple(range(1, len(shape) + 1)))
    for conv, ref in ((0, F.real + F.imag), (1, F.real - F.imag)):
        out = np.empty_like(x)
        f = Fuse()
        f.in_, f.out, f.scale = x.ctypes.data, out.ctypes.data, 1.0
        run(f, shape, dtype, conv, batch=2, fn=fast_fn(shape))
        err = np.max(np.abs(out - ref)) / np.max(np.abs(ref))
        assert err < (1e-12 if dtype == np.float64 else 2e-5)


FUSED_SHAPES = [((64,), "emu_hartley_fused"), ((16, 8), "emu_hartley_fused"), ((8, 4, 16), "emu_hartley_fused"),
                ((30,), "emu_hartley_fused"), ((10, 12), "emu_hartley_fused"), ((6, 5, 14), "emu_hartley_fused"),
                ((256,), "emu2_hartley_fused"), ((64, 128), "emu3_hartley_fused"), ((64, 64, 64), "emu3_hartley_fused"),
                ((64, 64, 1024), "emu3_hartley_fused"),  # last axis 1024 in fp64: the smallest final-pass tile
                ((2048, 64), "emu3_hartley_fused"), ((4096, 128), "emu3_hartley_fused"),  # two-level first-axis pass (64 x 32, 64 x 64)
                ((64, 4096), "emu3_hartley_fused")]  # 2-D VJP final pass on single line pairs (nk_final_single_2d: 4096 fp64)


@pytest.mark.parametrize("shape,fn", FUSED_SHAPES)
def test_fused_prologue_epilogue_emulation(shape, fn):
    """AMP_JVP prologue and VJP epilogue against their numpy definition."""
    rng = np.random.default_rng(1)
    n = int(np.prod(shape))
    nb = 7
    pidx = rng.integers(0, nb, size=shape).astype(np.int32)
    # bins symmetric under the sign flip of every single axis, like a real PowerSpace (|k| binning)
    idx = np.indices(shape)
    for d in range(len(shape)):
        flip = tuple((-idx[e]) % shape[e] if e == d else idx[e] for e in range(len(shape)))
        pidx = np.minimum(pidx, pidx[flip])
    pidx = pidx.astype(np.int32)
    amp, damp = rng.normal(size=nb), rng.normal(size=nb)
    xi, dxi, w, addend = (rng.normal(size=shape) for _ in range(4))
    H = lambda a: (lambda F: F.real + F.imag)(scipy.fft.fftn(a))  # noqa: E731
    out = np.empty(shape)
    f = Fuse()
    f.pro, f.in_, f.in2, f.pidx, f.amp, f.damp = 2, dxi.ctypes.data, xi.ctypes.data, pidx.ctypes.data, amp.ctypes.data, damp.ctypes.data
    f.epi, f.out, f.scale, f.offset = 0, out.ctypes.data, 0.5, 1.25
    run(f, shape, np.float64, fn=fn)
    ref = 0.5 * H(amp[pidx] * dxi + damp[pidx] * xi) + 1.25
    assert np.max(np.abs(out - ref)) < 1e-11 * np.max(np.abs(ref))
    # MUL prologue (in * in2)
    outm = np.empty(shape)
    f = Fuse()
    f.pro, f.in_, f.in2 = 3, dxi.ctypes.data, xi.ctypes.data
    f.epi, f.out, f.scale, f.offset = 0, outm.ctypes.data, 0.5, 0.0
    run(f, shape, np.float64, fn=fn)
    refm = 0.5 * H(dxi * xi)
    assert np.max(np.abs(outm - refm)) < 1e-11 * np.max(np.abs(refm))
    abar = np.zeros(nb)
    out2 = np.full(shape, 3.0)
    f = Fuse()
    f.pro, f.in_ = 0, w.ctypes.data
    f.epi, f.out, f.scale = 2, out2.ctypes.data, 0.25
    f.pidx, f.amp, f.xi, f.abar = pidx.ctypes.data, amp.ctypes.data, xi.ctypes.data, abar.ctypes.data
    f.addend, f.addend_scale, f.accumulate = addend.ctypes.data, 2.0, 1
    run(f, shape, np.float64, fn=fn)
    t = 0.25 * H(w)
    assert np.max(np.abs(out2 - (amp[pidx] * t + 2.0 * addend + 3.0))) < 1e-11 * np.max(np.abs(t))
    ref_abar = np.bincount(pidx.ravel(), weights=(xi * t).ravel(), minlength=nb)
    assert np.max(np.abs(abar - ref_abar)) < 1e-10 * max(1.0, np.max(np.abs(ref_abar)))
    # materialised amplitude field + T-typed da table + private accumulators give the same answer
    afield = amp[pidx].copy()
    out3 = np.empty(shape)
    f = Fuse()
    f.pro, f.in_, f.in2, f.pidx, f.amp, f.damp = 2, dxi.ctypes.data, xi.ctypes.data, pidx.ctypes.data, amp.ctypes.data, damp.ctypes.data
    f.afield, f.dampT = afield.ctypes.data, damp.ctypes.data
    f.epi, f.out, f.scale, f.offset = 0, out3.ctypes.data, 0.5, 1.25
    run(f, shape, np.float64, fn=fn)
    assert np.max(np.abs(out3 - ref)) < 1e-11 * np.max(np.abs(ref))
    priv = np.zeros(8 * 32)
    out4 = np.empty(shape)
    f = Fuse()
    f.pro, f.in_ = 0, w.ctypes.data
    f.epi, f.out, f.scale = 2, out4.ctypes.data, 0.25
    f.pidx, f.amp, f.xi, f.abar, f.afield = pidx.ctypes.data, amp.ctypes.data, xi.ctypes.data, priv.ctypes.data, afield.ctypes.data
    f.abar_copies, f.abar_stride = 8, 32
    run(f, shape, np.float64, fn=fn)
    assert np.max(np.abs(out4 - amp[pidx] * t)) < 1e-11 * np.max(np.abs(t))
    assert np.max(np.abs(priv.reshape(8, 32).sum(0)[:nb] - ref_abar)) < 1e-10 * max(1.0, np.max(np.abs(ref_abar)))
    if fn != "emu3_hartley_fused":
        return
    # OCTANT-shaped amplitude fields (nk_fuse.field_octant) + octant sums w8: strided-first pipeline only
    oct_sl = tuple(slice(0, s // 2 + 1) for s in shape)
    af8, daf8 = np.ascontiguousarray(amp[pidx][oct_sl]), np.ascontiguousarray(damp[pidx][oct_sl])
    out5 = np.empty(shape)
    f = Fuse()
    f.pro, f.in_, f.in2, f.pidx, f.amp, f.damp = 2, dxi.ctypes.data, xi.ctypes.data, pidx.ctypes.data, amp.ctypes.data, damp.ctypes.data
    f.afield, f.dafield, f.field_octant = af8.ctypes.data, daf8.ctypes.data, 1
    f.epi, f.out, f.scale, f.offset = 0, out5.ctypes.data, 0.5, 1.25
    run(f, shape, np.float64, fn=fn)
    assert np.max(np.abs(out5 - ref)) < 1e-11 * np.max(np.abs(ref))
    out6 = np.empty(shape)
    f = Fuse()
    f.pro, f.in_, f.pidx, f.amp = 1, xi.ctypes.data, pidx.ctypes.data, amp.ctypes.data
    f.afield, f.field_octant = af8.ctypes.data, 1
    f.epi, f.out, f.scale, f.offset = 0, out6.ctypes.data, 1.0, 0.0
    run(f, shape, np.float64, fn=fn)
    ref6 = H(amp[pidx] * xi)
    assert np.max(np.abs(out6 - ref6)) < 1e-11 * np.max(np.abs(ref6))
    w8 = np.full(af8.shape, np.nan)
    out7 = np.full(shape, 3.0)
    dummy = np.zeros(nb)
    f = Fuse()
    f.pro, f.in_ = 0, w.ctypes.data
    f.epi, f.out, f.scale = 2, out7.ctypes.data, 0.25
    f.pidx, f.amp, f.xi, f.abar = pidx.ctypes.data, amp.ctypes.data, xi.ctypes.data, dummy.ctypes.data
    f.afield, f.field_octant, f.w8 = af8.ctypes.data, 1, w8.ctypes.data
    f.addend, f.addend_scale, f.accumulate = addend.ctypes.data, 2.0, 1
    dq = np.zeros(1)
    f.value = dq.ctypes.data  # fused CG curvature: sum addend * out, taken in the same epilogue
    run(f, shape, np.float64, fn=fn)
    assert np.max(np.abs(out7 - (amp[pidx] * t + 2.0 * addend + 3.0))) < 1e-11 * np.max(np.abs(t))
    assert abs(dq[0] - np.sum(addend * out7)) < 1e-11 * np.sum(np.abs(addend * out7))
    got = np.bincount(pidx[oct_sl].ravel(), weights=w8.ravel(), minlength=nb)
    assert np.max(np.abs(got - ref_abar)) < 1e-10 * max(1.0, np.max(np.abs(ref_abar)))
    assert np.all(dummy == 0.0)


@pytest.mark.parametrize("shape,fn", [((8, 16), "emu_hartley_fused"), ((64, 64), "emu3_hartley_fused"),
                                      ((64, 64, 64), "emu3_hartley_fused")])
def test_likelihood_epilogue_emulation(shape, fn):
    rng = np.random.default_rng(2)
    xi = rng.normal(size=shape)
    d = rng.poisson(3.0, size=shape).astype(np.int64)
    gs, mid = np.empty(shape), np.empty(shape)
    val = np.zeros(1)
    f = Fuse()
    f.in_, f.epi, f.out, f.out2, f.scale, f.offset = xi.ctypes.data, 3, gs.ctypes.data, mid.ctypes.data, 0.1, 0.3
    f.lh_kind, f.nonlin, f.data, f.value = 1, 1, d.ctypes.data, val.ctypes.data
    run(f, shape, np.float64, fn=fn)
    F = scipy.fft.fftn(xi)
    s = 0.1 * (F.real + F.imag) + 0.3
    lam = np.exp(s)
    assert abs(val[0] - (lam.sum() - (d * s).sum())) < 1e-10 * abs(val[0])
    assert np.max(np.abs(gs - (lam - d))) < 1e-11 * np.max(np.abs(lam))
    assert np.max(np.abs(mid - lam)) < 1e-11 * np.max(np.abs(lam))


def run_sandwich(f, shape, dtype, scale_first, conv=0, batch=1, cx=0):
    shp = (ctypes.c_int64 * len(shape))(*shape)
    fn = lib().emu4_hartley_sandwich
    fn.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_int64, ctypes.c_void_p,
                   ctypes.c_double, ctypes.c_int, ctypes.c_int]
    rc = fn(len(shape), shp, 0 if dtype == np.float32 else 1, batch, ctypes.addressof(f), scale_first, conv, cx)
    assert rc == 0, rc


@pytest.mark.parametrize("shape", [(64, 128), (128, 128), (64, 256), (64, 64, 128), (128, 64, 128), (64, 128, 256)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("cx", [0, 1, 2, 3, 8])
def test_sandwich_emulation(shape, dtype, cx):
    """nk_hartley_sandwich (five-pass H D H, nk_fft3.h) against two scipy transforms: plain, both conventions, a
    batch of 2, constant and field diagonal.  cx bit 0: complex-plane exchange of the fused pass, bit 1: persistent
    workgroups (5 of them) with register prefetch of the next tile, 8: composed twiddles."""
    rng = np.random.default_rng(5)
    x = rng.normal(size=(2,) + shape).astype(dtype)
    m = rng.normal(size=(2,) + shape).astype(dtype)
    axes = tuple(range(1, len(shape) + 1))
    tol = 1e-12 if dtype == np.float64 else 5e-5
    for conv, sg in ((0, 1.0), (1, -1.0)):
        H = lambda a: (lambda F: F.real + sg * F.imag)(scipy.fft.fftn(a.astype(np.float64), axes=axes))  # noqa: E731
        for mul in (None, m):
            out = np.empty_like(x)
            f = Fuse()
            f.in_, f.out, f.scale, f.offset, f.mul_scalar = x.ctypes.data, out.ctypes.data, 0.5, 0.75, 1.5
            if mul is not None:
                f.mul = mul.ctypes.data
            run_sandwich(f, shape, dtype, 0.25, conv, batch=2, cx=cx)
            mid = 1.5 * 0.25 * H(x) * (1.0 if mul is None else mul.astype(np.float64))
            ref = 0.5 * H(mid) + 0.75
            err = np.max(np.abs(out - ref)) / np.max(np.abs(ref))
            assert err < tol, (conv, mul is not None, err)


@pytest.mark.parametrize("shape", [(64, 128), (64, 64, 128), (64, 64, 1024)])
def test_sandwich_fused_emulation(shape):
    """The metric application through the sandwich: AMP_JVP prologue with octant fields, field diagonal, VJP epilogue
    with octant sums, addend, accumulation and the fused curvature dot."""
    rng = np.random.default_rng(6)
    nb = 7
    pidx = rng.integers(0, nb, size=shape).astype(np.int32)
    idx = np.indices(shape)
    for d in range(len(shape)):
        flip = tuple((-idx[e]) % shape[e] if e == d else idx[e] for e in range(len(shape)))
        pidx = np.minimum(pidx, pidx[flip])
    pidx = pidx.astype(np.int32)
    amp, damp = rng.normal(size=nb), rng.normal(size=nb)
    xi, dxi, addend, mid = (rng.normal(size=shape) for _ in range(4))
    H = lambda a: (lambda F: F.real + F.imag)(scipy.fft.fftn(a))  # noqa: E731
    oct_sl = tuple(slice(0, s // 2 + 1) for s in shape)
    af8, daf8 = np.ascontiguousarray(amp[pidx][oct_sl]), np.ascontiguousarray(damp[pidx][oct_sl])
    pidx8 = np.ascontiguousarray(pidx[oct_sl])
    for use_mid in (False, True):
        for octant in (True, False):
            w8 = np.full(af8.shape, np.nan)
            out = np.full(shape, 3.0)
            abar = np.zeros(nb)
            dq = np.zeros(1)
            f = Fuse()
            f.pro, f.in_, f.in2, f.pidx, f.amp, f.damp = 2, dxi.ctypes.data, xi.ctypes.data, pidx.ctypes.data, amp.ctypes.data, damp.ctypes.data
            f.mul_scalar = 0.7
            if use_mid:
                f.mul = mid.ctypes.data
            f.epi, f.out, f.scale = 2, out.ctypes.data, 0.25
            f.xi, f.abar = xi.ctypes.data, abar.ctypes.data
            f.addend, f.addend_scale, f.accumulate = addend.ctypes.data, 2.0, 1
            if octant:
                f.afield, f.field_octant, f.w8 = af8.ctypes.data, 1, w8.ctypes.data
                if use_mid:  # da gathered from its table through the octant bin index instead of an expanded field
                    f.pidx_octant, f.dampT = pidx8.ctypes.data, damp.ctypes.data
                else:
                    f.dafield = daf8.ctypes.data
                f.value = dq.ctypes.data
            run_sandwich(f, shape, np.float64, 0.5)
            s = 0.5 * H(amp[pidx] * dxi + damp[pidx] * xi)
            t = 0.25 * H(0.7 * (mid if use_mid else 1.0) * s)
            ref = amp[pidx] * t + 2.0 * addend + 3.0
            assert np.max(np.abs(out - ref)) < 1e-11 * np.max(np.abs(t)), (use_mid, octant)
            ref_abar = np.bincount(pidx.ravel(), weights=(xi * t).ravel(), minlength=nb)
            if octant:
                got = np.bincount(pidx[oct_sl].ravel(), weights=w8.ravel(), minlength=nb)
                assert abs(dq[0] - np.sum(addend * out)) < 1e-11 * np.sum(np.abs(addend * out))
            else:
                got = abar
            assert np.max(np.abs(got - ref_abar)) < 1e-10 * max(1.0, np.max(np.abs(ref_abar)))


@pytest.mark.parametrize("shape,dtype", [((64, 64, 128), np.float64), ((64, 64, 128), np.float32), ((12, 10), np.float64),
                                         ((64, 128), np.float32)])
def test_vjp_carry_chain_is_a_chain_of_plain_additions(shape, dtype):
    """nk_fuse.carry1 / carry2 / accumulate: out = out + (carry2 + (carry1 + g)) with the sample's own contribution g
    rounded first -- BIT-identical to storing g and adding the partial sums one by one (what another rank would do):
    the pairwise sum over samples (utilities.py:349-414) inside the epilogues.  Register-resident final pass (octant
    fields, compile-time MODE 6 / 14 and the run-time path) and the generic kernels."""
    rng = np.random.default_rng(11)
    nb = 5
    pidx = rng.integers(0, nb, size=shape).astype(np.int32)
    idx = np.indices(shape)
    for d in range(len(shape)):
        flip = tuple((-idx[e]) % shape[e] if e == d else idx[e] for e in range(len(shape)))
        pidx = np.minimum(pidx, pidx[flip])
    pidx = pidx.astype(np.int32)
    amp = rng.normal(size=nb)
    xi, w, addend, c1, c2, run0 = (rng.normal(size=shape).astype(dtype) for _ in range(6))
    oct_sl = tuple(slice(0, s // 2 + 1) for s in shape)
    af8 = np.ascontiguousarray(amp[pidx][oct_sl]).astype(dtype)
    fast = all(n & (n - 1) == 0 for n in shape)

    def vjp(out, accumulate, carries, with_addend):
        w8 = np.zeros(af8.shape)
        abar = np.zeros(nb)
        f = Fuse()
        f.pro, f.in_ = 0, w.ctypes.data
        f.epi, f.out, f.scale = 2, out.ctypes.data, 0.25
        f.pidx, f.amp, f.xi, f.abar = pidx.ctypes.data, amp.ctypes.data, xi.ctypes.data, abar.ctypes.data
        if with_addend:
            f.addend, f.addend_scale = addend.ctypes.data, 0.5
        f.accumulate = 1 if accumulate else 0
        f.carry1, f.carry2 = [c.ctypes.data for c in carries] + [None] * (2 - len(carries))
        if fast:
            f.afield, f.field_octant, f.w8 = af8.ctypes.data, 1, w8.ctypes.data
        run(f, shape, dtype, fn=fast_fn(shape) if fast else "emu_hartley_fused")
        return out

    for with_addend in (False, True):
        g = vjp(np.full(shape, np.nan, dtype=dtype), False, (), with_addend)  # the sample alone
        for carries, accumulate in (((c1,), True), ((c1, c2), True), ((c1,), False), ((), True)):
            want = g.copy()
            for c in carries:
                want = c + want
            if accumulate:
                want = run0 + want
            got = vjp(run0.copy(), accumulate, carries, with_addend)
            assert np.array_equal(got, want), (with_addend, len(carries), accumulate)


def test_wide_schedule_emulation():
    """The in-place strided pass of 1024-point fp32 lines runs 256 threads x 64 elements (radix 64 x 16, SchedW): first
    axis of a 3-D strided-first transform and the middle-axis passes of a sandwich."""
    rng = np.random.default_rng(8)
    shape = (1024, 64, 128)
    x = rng.normal(size=shape).astype(np.float32)
    F = scipy.fft.fftn(x.astype(np.float64))
    ref = F.real + F.imag
    out = np.empty_like(x)
    f = Fuse()
    f.in_, f.out, f.scale = x.ctypes.data, out.ctypes.data, 1.0
    run(f, shape, np.float32, fn="emu3_hartley_fused")
    assert np.max(np.abs(out - ref)) / np.max(np.abs(ref)) < 2e-5
    shape = (64, 1024, 128)
    x = rng.normal(size=shape).astype(np.float32)
    H = lambda a: (lambda F: F.real + F.imag)(scipy.fft.fftn(a.astype(np.float64)))  # noqa: E731
    out = np.empty_like(x)
    f = Fuse()
    f.in_, f.out, f.scale, f.mul_scalar = x.ctypes.data, out.ctypes.data, 1.0 / x.size, 1.0
    run_sandwich(f, shape, np.float32, 1.0)
    ref = H(H(x)) / x.size
    assert np.max(np.abs(out - ref)) / np.max(np.abs(ref)) < 5e-5
    # ... and the fused first-axis pass itself on the wide schedule (half-column complex exchange), field diagonal
    shape = (1024, 128)
    x = rng.normal(size=shape).astype(np.float32)
    m = rng.normal(size=shape).astype(np.float32)
    for mode in (4, 6, 8):
        out = np.empty_like(x)
        f = Fuse()
        f.in_, f.out, f.scale, f.mul_scalar, f.mul = x.ctypes.data, out.ctypes.data, 1.0 / x.size, 1.0, m.ctypes.data
        run_sandwich(f, shape, np.float32, 1.0, cx=mode)
        ref = H(m.astype(np.float64) * H(x)) / x.size
        assert np.max(np.abs(out - ref)) / np.max(np.abs(ref)) < 5e-5


def test_sandwich_cg_direction_emulation():
    """The pending CG direction update d <- beta d + r rides in the sandwich's first pass (written back to d)."""
    rng = np.random.default_rng(9)
    shape = (64, 64, 128)
    nb = 7
    pidx = rng.integers(0, nb, size=shape).astype(np.int32)
    idx = np.indices(shape)
    for d_ in range(3):
        flip = tuple((-idx[e]) % shape[e] if e == d_ else idx[e] for e in range(3))
        pidx = np.minimum(pidx, pidx[flip])
    pidx = pidx.astype(np.int32)
    amp, damp = rng.normal(size=nb), rng.normal(size=nb)
    xi, d, r = (rng.normal(size=shape) for _ in range(3))
    scal = np.array([2.0, 0.0, 0.5, 0, 0, 0, 0, 0])
    beta = 0.25
    H = lambda a: (lambda F: F.real + F.imag)(scipy.fft.fftn(a))  # noqa: E731
    oct_sl = tuple(slice(0, s_ // 2 + 1) for s_ in shape)
    af8, daf8 = np.ascontiguousarray(amp[pidx][oct_sl]), np.ascontiguousarray(damp[pidx][oct_sl])
    d_new = beta * d + r
    d_work = d.copy()
    out = np.empty(shape)
    abar, w8 = np.zeros(nb), np.full(af8.shape, np.nan)
    f = Fuse()
    f.pro, f.in_, f.in2, f.pidx, f.amp, f.damp = 2, d_work.ctypes.data, xi.ctypes.data, pidx.ctypes.data, amp.ctypes.data, damp.ctypes.data
    f.afield, f.dafield, f.field_octant, f.w8 = af8.ctypes.data, daf8.ctypes.data, 1, w8.ctypes.data
    f.cg_r, f.cg_scal = r.ctypes.data, scal.ctypes.data
    f.mul_scalar, f.epi, f.out, f.scale = 1.0, 2, out.ctypes.data, 0.25
    f.xi, f.abar, f.addend, f.addend_scale = xi.ctypes.data, abar.ctypes.data, d_work.ctypes.data, 1.0
    run_sandwich(f, shape, np.float64, 0.5)
    assert np.array_equal(d_work, d_new)
    t = 0.25 * H(0.5 * H(amp[pidx] * d_new + damp[pidx] * xi))
    assert np.max(np.abs(out - (amp[pidx] * t + d_new))) < 1e-11 * np.max(np.abs(t))


@pytest.mark.parametrize("shape", [(64, 128), (64, 64, 64), (64, 64, 128)])
@pytest.mark.parametrize("kind", ["gauss", "gauss_icov", "poisson"])
def test_io32_wide_forward_emulation(shape, kind):
    """nk_fuse.io32: float excitations / data / outputs at both ends of an fp64 transform (prologue class 9, epilogue
    class 5) -- the value / gradient forward transform of a model with fp32 fields, against its numpy definition in fp64
    (the reference promotes fp32 xi at the product with the fp64 amplitude, library/correlated_fields.py:755-764)."""
    rng = np.random.default_rng(11)
    nb = 7
    pidx = rng.integers(0, nb, size=shape).astype(np.int32)
    idx = np.indices(shape)
    for d in range(len(shape)):
        flip = tuple((-idx[e]) % shape[e] if e == d else idx[e] for e in range(len(shape)))
        pidx = np.minimum(pidx, pidx[flip])
    pidx = pidx.astype(np.int32)
    amp = 0.3 * rng.normal(size=nb) / np.sqrt(np.prod(shape))
    xi = rng.normal(size=shape).astype(np.float32)
    H = lambda a: (lambda F: F.real + F.imag)(scipy.fft.fftn(a))  # noqa: E731
    oct_sl = tuple(slice(0, s // 2 + 1) for s in shape)
    af8 = np.ascontiguousarray(amp[pidx][oct_sl])  # fp64 octant field
    s = 0.5 * H(amp[pidx] * xi.astype(np.float64)) + 1.25
    value = np.zeros(1)
    out, out2 = np.full(shape, np.nan, dtype=np.float32), np.full(shape, np.nan, dtype=np.float32)
    f = Fuse()
    f.pro, f.in_, f.pidx, f.amp = 1, xi.ctypes.data, pidx.ctypes.data, amp.ctypes.data
    f.afield, f.field_octant, f.io32 = af8.ctypes.data, 1, 1
    f.epi, f.out, f.out2, f.scale, f.offset, f.value = 3, out.ctypes.data, out2.ctypes.data, 0.5, 1.25, value.ctypes.data
    if kind == "poisson":
        data = rng.poisson(np.exp(s)).astype(np.int64)
        f.lh_kind, f.nonlin, f.data = 1, 1, data.ctypes.data
        lam = np.exp(s)
        ref_e, ref_gs, ref_w = np.sum(lam - data * s), lam - data, lam
    else:
        data = (s + 0.1 * rng.normal(size=shape)).astype(np.float32)
        f.lh_kind, f.nonlin, f.data, f.icov_scalar = 0, 0, data.ctypes.data, 100.0
        ic = 100.0
        if kind == "gauss_icov":
            icf = rng.uniform(50.0, 150.0, size=shape).astype(np.float32)
            f.icov = icf.ctypes.data
            ic = icf.astype(np.float64)
        r = s - data.astype(np.float64)
        ref_e, ref_gs, ref_w = 0.5 * np.sum(ic * r * r), ic * r, ic * np.ones(shape)
    run(f, shape, np.float64, fn="emu3_hartley_fused")
    assert abs(value[0] - ref_e) < 1e-12 * abs(ref_e)
    # the outputs are the fp64 results rounded ONCE to float
    assert np.array_equal(out, ref_gs.astype(np.float32)) or np.max(np.abs(out - ref_gs)) < 1.3e-7 * np.max(np.abs(ref_gs))
    assert np.max(np.abs(out2 - ref_w)) < 1.3e-7 * np.max(np.abs(ref_w))


def test_wide_forward_brings_the_fp32_gradient_to_the_fp64_oracle():
    """Value + gradient of the correlated-field model with fp32 fields through the emulated kernels: fp64 forward
    transform with float arrays at both ends (io32), fp32 adjoint transform -- against the fp64 oracle on IDENTICAL
    (fp32-rounded) inputs.  The all-fp32 forward leaves a coherent gain error of ~6e-8 on the signal, which the residual
    N^-1 (s - d) amplifies (1e-5 .. 1e-3 of the gradient); with the wide forward every key is at the 1e-7 level."""
    from oracle import nifty_oracle as orc

    shape = (64, 64, 128)
    cf = orc.CFModel(shape, None, orc.CFParams(offset_mean=2.0))
    rng = np.random.default_rng(22)
    x = {k: np.asarray(a) for k, a in cf.draw_latent(rng).items()}
    data = cf.forward(x) + 0.1 * rng.normal(size=shape)
    x["xi"] = x["xi"].astype(np.float32).astype(np.float64)
    data32 = data.astype(np.float32)
    lh = orc.Likelihood("gaussian", data32.astype(np.float64), icov=100.0)
    val, grad = orc.Linearized(cf, lh, x).value_grad()
    st = cf.amplitude_state(x)
    pin = cf.geo.pindex.astype(np.int32)
    oct_sl = tuple(slice(0, s // 2 + 1) for s in shape)
    a64 = st["a"][pin]
    xi32 = x["xi"].astype(np.float32)

    def gradient(wide):
        gs, value = np.empty(shape, dtype=np.float32), np.zeros(1)
        f = Fuse()
        f.pro, f.pidx, f.amp, f.field_octant = 1, pin.ctypes.data, st["a"].ctypes.data, 1
        f.epi, f.out, f.scale, f.offset, f.value = 3, gs.ctypes.data, 1.0, 2.0, value.ctypes.data
        f.lh_kind, f.nonlin, f.data, f.icov_scalar = 0, 0, data32.ctypes.data, 100.0
        af = np.ascontiguousarray(a64[oct_sl] if wide else a64[oct_sl].astype(np.float32))
        f.in_, f.afield, f.io32 = xi32.ctypes.data, af.ctypes.data, int(wide)
        run(f, shape, np.float64 if wide else np.float32, fn="emu3_hartley_fused")
        # adjoint: fp32 transform with the VJP epilogue (octant sums)
        af32 = np.ascontiguousarray(a64[oct_sl].astype(np.float32))
        gxi, w8, dummy = np.empty(shape, dtype=np.float32), np.zeros(af32.shape), np.zeros(cf.geo.nb)
        f = Fuse()
        f.pro, f.in_ = 0, gs.ctypes.data
        f.epi, f.out, f.scale = 2, gxi.ctypes.data, 1.0
        f.pidx, f.amp, f.xi, f.abar = pin.ctypes.data, st["a"].ctypes.data, xi32.ctypes.data, dummy.ctypes.data
        f.afield, f.field_octant, f.w8 = af32.ctypes.data, 1, w8.ctypes.data
        f.addend, f.addend_scale = xi32.ctypes.data, 1.0
        run(f, shape, np.float32, fn="emu3_hartley_fused")
        abar = np.bincount(pin[oct_sl].ravel(), weights=w8.ravel(), minlength=cf.geo.nb)
        out = cf.amplitude_vjp(st, abar)
        out = {k: out[k] + x[k] for k in out}
        out["xi"] = gxi.astype(np.float64)
        prior = 0.5 * sum(float(np.sum(np.asarray(x[k]) ** 2)) for k in x)
        return value[0] + prior, out

    scale = max(float(np.max(np.abs(grad[k]))) for k in grad)

    def worst(g):
        return max(float(np.max(np.abs(g[k] - grad[k]))) / scale for k in grad)

    v32, g32 = gradient(False)
    v64, g64 = gradient(True)
    assert abs(v64 - val) < 1e-12 * abs(val)
    assert worst(g64) < 1e-6, worst(g64)
    assert worst(g32) > 3 * worst(g64)  # what the wide forward removes
    per_key = max(float(np.max(np.abs(g64[k] - grad[k]))) / float(np.max(np.abs(grad[k]))) for k in grad)
    assert per_key < 1e-5, per_key
