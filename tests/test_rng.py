"""numpy's Generator(PCG64).normal stream on the device (nifty_amd/csrc/nk_rng.h, nk_rng.hip).

The checker is numpy itself -- the reference's third-party RNG (nifty/cl/random.py:146-237), present on every box.
CPU tests run the host emulation of the per-chunk kernel bodies (tests/emu/emu_rng.cpp: the same functions the kernels
call) and the Python side of the state hand-over; GPU tests run nk_pcg64_normal through the C ABI.
Bar: bit-identical values and generator state.  On the device the |x| > 3.654 tail (2.7e-4 of the draws) goes through the
device math library's log1p and may differ from the host libm in the last bits (tolerance 4 ulp, written below)."""
import ctypes
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu", "libnk_emu.so")
U64P = ctypes.POINTER(ctypes.c_uint64)
TAIL = 3.6541528853610087


def _words(rng):
    s = rng.bit_generator.state["state"]
    m = 2**64 - 1
    return (np.array([s["state"] >> 64, s["state"] & m], dtype=np.uint64), np.array([s["inc"] >> 64, s["inc"] & m], dtype=np.uint64))


def _emu():
    if not os.path.exists(EMU):
        pytest.skip("emulation library not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(EMU)
    lib.emu_rng_chunks_for.restype = ctypes.c_int64
    lib.emu_pcg64_normal_chunked.restype = ctypes.c_uint
    return lib


CASES = [(42, 2_000_000, 0.0, 1.0), (7, 1, 0.0, 1.0), (8, 63, 1.5, 0.1), (9, 100_003, -2.0, 3.0), (31, 64, 0.0, 1.0),
         (10, 3_000_000, 0.0, 10.0)]


@pytest.mark.parametrize("seed,n,mean,std", CASES)
def test_serial_restatement_equals_numpy(seed, n, mean, std):
    lib = _emu()
    rng = np.random.default_rng(np.random.SeedSequence(seed))
    state, inc = _words(rng)
    ref = rng.normal(mean, std, n)
    after, _ = _words(rng)
    out = np.empty(n)
    used = ctypes.c_uint64(0)
    lib.emu_pcg64_normal_serial(state.ctypes.data_as(U64P), inc.ctypes.data_as(U64P), ctypes.c_int64(n), ctypes.c_double(mean),
                                ctypes.c_double(std), out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.byref(used))
    assert np.array_equal(out.view(np.uint64), ref.view(np.uint64))
    adv = np.zeros(2, dtype=np.uint64)
    lib.emu_pcg64_advance(state.ctypes.data_as(U64P), inc.ctypes.data_as(U64P), ctypes.c_uint64(used.value), adv.ctypes.data_as(U64P))
    assert np.array_equal(adv, after)
    if n >= 2_000_000:  # the sample exercises the wedge and the tail branch
        assert (np.abs((ref - mean) / std) > TAIL).sum() > 100
        assert used.value > n


@pytest.mark.parametrize("seed,n,mean,std", CASES)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_chunked_kernel_bodies_equal_numpy(seed, n, mean, std, dtype):
    lib = _emu()
    rng = np.random.default_rng(np.random.SeedSequence(seed))
    state, inc = _words(rng)
    ref = rng.normal(mean, std, n).astype(dtype)
    after, _ = _words(rng)
    nch = lib.emu_rng_chunks_for(ctypes.c_int64(n), 0)
    out = np.empty(n, dtype=dtype)
    used = ctypes.c_uint64(0)
    err = lib.emu_pcg64_normal_chunked(state.ctypes.data_as(U64P), inc.ctypes.data_as(U64P), ctypes.c_int64(n), ctypes.c_double(mean),
                                       ctypes.c_double(std), 1 if dtype == np.float64 else 0, out.ctypes.data_as(ctypes.c_void_p),
                                       ctypes.byref(used), ctypes.c_int64(nch))
    assert err == 0
    assert np.array_equal(out.view(np.uint64 if dtype == np.float64 else np.uint32), ref.view(np.uint64 if dtype == np.float64 else np.uint32))
    adv = np.zeros(2, dtype=np.uint64)
    lib.emu_pcg64_advance(state.ctypes.data_as(U64P), inc.ctypes.data_as(U64P), ctypes.c_uint64(used.value), adv.ctypes.data_as(U64P))
    assert np.array_equal(adv, after)


def test_too_few_chunks_is_reported_not_wrong():
    lib = _emu()
    rng = np.random.default_rng(5)
    state, inc = _words(rng)
    n = 10_000
    out = np.empty(n)
    used = ctypes.c_uint64(0)
    err = lib.emu_pcg64_normal_chunked(state.ctypes.data_as(U64P), inc.ctypes.data_as(U64P), ctypes.c_int64(n), ctypes.c_double(0.0),
                                       ctypes.c_double(1.0), 1, out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(used),
                                       ctypes.c_int64(n // 64))  # 64 draws per chunk hold < 64 normals
    assert err == 2


@pytest.mark.parametrize("delta", [0, 1, 63, 64, 12345678901, 2**63 + 17])
def test_python_jump_ahead_equals_numpy_advance(delta):
    from nifty_amd.backend import _pcg64_advanced

    rng = np.random.default_rng(123)
    st = rng.bit_generator.state["state"]
    got = _pcg64_advanced(int(st["state"]), int(st["inc"]), delta)
    rng.bit_generator.advance(delta)
    assert got == int(rng.bit_generator.state["state"]["state"])
    lib = _emu()
    rng2 = np.random.default_rng(123)
    state, inc = _words(rng2)
    adv = np.zeros(2, dtype=np.uint64)
    lib.emu_pcg64_advance(state.ctypes.data_as(U64P), inc.ctypes.data_as(U64P), ctypes.c_uint64(delta), adv.ctypes.data_as(U64P))
    assert (int(adv[0]) << 64) | int(adv[1]) == got


def test_tables_header_matches_installed_numpy():
    """The committed bit patterns are numpy's: a single-draw generator built from them reproduces numpy for every strip."""
    lib = _emu()
    rng = np.random.default_rng(99)
    n = 400_000  # every one of the 256 strips is hit > 1000 times
    state, inc = _words(rng)
    raw = np.random.default_rng(99).bit_generator.random_raw(n)
    assert len(np.unique(raw & 0xFF)) == 256
    ref = rng.normal(size=n)
    out = np.empty(n)
    used = ctypes.c_uint64(0)
    lib.emu_pcg64_normal_serial(state.ctypes.data_as(U64P), inc.ctypes.data_as(U64P), ctypes.c_int64(n), ctypes.c_double(0.0),
                                ctypes.c_double(1.0), out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.byref(used))
    assert np.array_equal(out, ref)


def test_small_and_host_forced_draws_take_the_host_path():
    from nifty_amd import config, random

    with random.Context(11):
        a = random.Random.normal_on_device(np.float64, (100,), 0.0, 1.0, torch.device("cpu"))
    with random.Context(11):
        b = random.Random.normal(np.float64, (100,))
    assert np.array_equal(a.numpy(), b)
    with random.Context(13):  # a host target never reaches the device kernel, whatever the size
        a = random.Random.normal_on_device(np.float64, (1 << 16,), 0.0, 1.0, torch.device("cpu"))
    with random.Context(13):
        b = random.Random.normal(np.float64, (1 << 16,))
    assert np.array_equal(a.numpy(), b)
    config.update("sampling_rng", "numpy_host")
    try:
        with random.Context(12):
            a = random.Random.normal_on_device(np.float32, (1 << 16,), 1.0, 2.0, torch.device("cpu"))
        with random.Context(12):
            b = random.Random.normal(np.float32, (1 << 16,), 1.0, 2.0)
        assert np.array_equal(a.numpy(), b)
    finally:
        config.update("sampling_rng", "numpy")
    with pytest.raises(ValueError):
        config.update("sampling_rng", "nonsense")


# ---- device ---------------------------------------------------------------------------------------------------------
def _ulps(a, b):
    ia = a.view(np.int64 if a.dtype == np.float64 else np.int32).astype(np.int64)
    ib = b.view(np.int64 if b.dtype == np.float64 else np.int32).astype(np.int64)
    return np.abs(ia - ib)


def _check_device_draw(got, ref64, mean, std, dtype):
    ref = ref64.astype(dtype)
    tail = np.abs((ref64 - mean) / std) > TAIL
    assert np.array_equal(got[~tail], ref[~tail])  # bit-identical outside the tail
    if tail.any():
        assert _ulps(got[tail], ref[tail]).max() <= 4  # log1p: device math library vs host libm
    return int(tail.sum()), int((got != ref).sum())


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,mean,std", CASES + [(77, (1 << 24) + 12345, 0.0, 1.0)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_device_normal_equals_numpy(seed, n, mean, std, dtype):
    from nifty_amd import backend as B

    rng = np.random.default_rng(np.random.SeedSequence(seed))
    ref_rng = np.random.default_rng(np.random.SeedSequence(seed))
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    got = B.pcg64_normal(rng, mean, std, (n,), tdt, torch.device("cuda:0")).cpu().numpy()
    ref = ref_rng.normal(mean, std, n)
    _check_device_draw(got, ref, mean, std, dtype)
    # the host generator continues exactly where numpy's own draw would have left it
    assert rng.bit_generator.state == ref_rng.bit_generator.state
    assert np.array_equal(rng.normal(size=5), ref_rng.normal(size=5))
    assert np.array_equal(rng.integers(0, 2, size=9), ref_rng.integers(0, 2, size=9))


@pytest.mark.gpu
def test_device_normal_keeps_the_cached_uint32():
    """A pending 32-bit half draw (Generator.integers) survives the device draw like it survives numpy's own normal."""
    from nifty_amd import backend as B

    rng, ref_rng = np.random.default_rng(3), np.random.default_rng(3)
    for g in (rng, ref_rng):
        g.integers(0, 2, size=3, dtype=np.uint32)
    assert rng.bit_generator.state["has_uint32"] == ref_rng.bit_generator.state["has_uint32"]
    got = B.pcg64_normal(rng, 0.0, 1.0, (70000,), torch.float64, torch.device("cuda:0")).cpu().numpy()
    ref = ref_rng.normal(size=70000)
    _check_device_draw(got, ref, 0.0, 1.0, np.float64)
    assert rng.bit_generator.state == ref_rng.bit_generator.state
    assert np.array_equal(rng.integers(0, 1 << 20, size=7, dtype=np.uint32), ref_rng.integers(0, 1 << 20, size=7, dtype=np.uint32))


@pytest.mark.gpu
def test_device_fields_draw_the_reference_stream():
    """Field.from_random / MultiField.from_random on a device: same values and same generator hand-over as on the host."""
    import nifty_amd as ift
    from nifty_amd import random

    dom = ift.RGSpace((96, 1024))
    md = ift.MultiDomain.make({"a": ift.RGSpace(7), "b": dom, "c": ift.RGSpace(3)})
    for dtype in (np.float64, np.float32, np.complex128):
        with random.Context(21):
            host = ift.from_random(md, "normal", dtype=dtype, std=2.0, mean=0.5)
            host_next = random.current_rng().normal(size=3)
        with random.Context(21):
            dev = ift.from_random(md, "normal", dtype=dtype, device_id=0, std=2.0, mean=0.5)
            dev_next = random.current_rng().normal(size=3)
        for k in md.keys():
            h, d = host[k].asnumpy(), dev[k].asnumpy()
            assert d.dtype == h.dtype and dev[k].device_id == 0
            if np.iscomplexobj(h):
                same = (h == d)
            else:
                same = (h == d)
            assert same.mean() > 0.999 and np.allclose(h, d, rtol=1e-14 if dtype != np.float32 else 1e-6, atol=0)
        assert np.array_equal(host_next, dev_next)


@pytest.mark.gpu
def test_engine_draws_equal_host_draws():
    """FusedModel.draw_prior / draw_lh_noise with the stream computed on the device vs numpy_host."""
    from nifty_amd import config, random
    from nifty_amd.engine import FusedModel

    model = FusedModel((64, 64, 128), offset_mean=2.0, likelihood="gaussian", icov=100.0, dtype=torch.float32, device="cuda:0")
    res = {}
    for mode in ("numpy", "numpy_host"):
        config.update("sampling_rng", mode)
        try:
            with random.Context(5):
                x = model.draw_prior()
                model.set_data(model.signal(x), 100.0)
                lp = model.linearize(x)
                nj = model.draw_lh_noise(lp)
                res[mode] = (x.xi.cpu().numpy(), x.small.cpu().numpy(), nj.xi.cpu().numpy(), random.current_rng().normal(size=2))
        finally:
            config.update("sampling_rng", "numpy")
    a, b = res["numpy"], res["numpy_host"]
    assert (a[0] != b[0]).mean() < 1e-3 and np.allclose(a[0], b[0], rtol=1e-6, atol=0)
    assert np.array_equal(a[1], b[1])
    assert np.allclose(a[2], b[2], rtol=1e-4, atol=1e-4 * np.abs(b[2]).max())
    assert np.array_equal(a[3], b[3])


@pytest.mark.gpu
def test_device_normal_full_size_stream():
    """BASELINE size: 1024^3 fp32 normals, compared with numpy slice by slice (the stream does not depend on how it is cut)."""
    from nifty_amd import backend as B

    n = 1 << 30
    rng = np.random.default_rng(np.random.SeedSequence(42))
    ref_rng = np.random.default_rng(np.random.SeedSequence(42))
    got = B.pcg64_normal(rng, 0.0, 1.0, (n,), torch.float32, torch.device("cuda:0"))
    step = 1 << 26
    ntail = ndiff = 0
    for lo in range(0, n, step):
        ref = ref_rng.normal(0.0, 1.0, step)
        t, d = _check_device_draw(got[lo:lo + step].cpu().numpy(), ref, 0.0, 1.0, np.float32)
        ntail += t
        ndiff += d
    assert rng.bit_generator.state == ref_rng.bit_generator.state
    assert 2.0e-4 * n < ntail < 3.5e-4 * n
    assert ndiff <= ntail


# ---- fixed-rate draws: uniform and pm1 (nk_pcg64_uniform / nk_pcg64_pm1) ------------------------------------------------------
def _fixed_emu(rng, n, mode, dt, low=0.0, high=1.0):
    lib = _emu()
    s, i = _words(rng)
    out = np.empty(2 * n if mode == 2 else n, dtype=dt)
    lib.emu_pcg64_fixed(s.ctypes.data_as(U64P), i.ctypes.data_as(U64P), ctypes.c_int64(n), ctypes.c_double(low), ctypes.c_double(high), mode,
                        0 if dt == np.float32 else 1, out.ctypes.data_as(ctypes.c_void_p))
    return out


@pytest.mark.parametrize("n", [1, 5, 63, 64, 65, 1000, 4097])
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_fixed_rate_bodies_equal_numpy(n, dt):
    """The per-thread body of k_rng_fixed (host emulation) against numpy itself, bit for bit: `uniform`, the +-1 draws of
    `2 * integers(0, 2) - 1` and the complex units of `integers(0, 4)` (reference random.py:239-258)."""
    r = np.random.default_rng(n)
    got = _fixed_emu(r, n, 0, dt, 1.5, 4.0)
    assert np.array_equal(got, r.uniform(1.5, 4.0, n).astype(dt))
    r = np.random.default_rng(n + 7)
    got = _fixed_emu(r, n, 1, dt)
    assert np.array_equal(got, (2 * r.integers(0, 2, size=n) - 1).astype(dt))
    r = np.random.default_rng(n + 9)
    got = _fixed_emu(r, n, 2, dt)
    units = np.array([1, 1j, -1, -1j])
    assert np.array_equal(got[0::2] + 1j * got[1::2], units[r.integers(0, 4, size=n)])


@pytest.mark.gpu
def test_uniform_and_pm1_on_the_device_equal_numpy_including_generator_state():
    """Field.from_random(random_type="uniform" | "pm1", device_id=0) draws the reference's numpy stream on the device: the
    values, AND the generator afterwards (raw position, buffered 32-bit half word) -- checked by the draws that follow."""
    import nifty_amd as ift

    n = 70001  # odd: a pm1 draw leaves half a raw value buffered
    dom = ift.UnstructuredDomain(n)
    units = np.array([1, 1j, -1, -1j])
    for dt in (np.float64, np.float32):
        cdt = np.complex128 if dt == np.float64 else np.complex64
        ift.random.push_sseq_from_seed(11)
        ref_rng = np.random.default_rng(np.random.SeedSequence(11))
        try:
            u = ift.from_random(dom, "uniform", dtype=dt, device_id=0, low=-2.0, high=3.0)
            assert u.device_id == 0 and np.array_equal(u.asnumpy(), ref_rng.uniform(-2.0, 3.0, n).astype(dt))
            p = ift.from_random(dom, "pm1", dtype=dt, device_id=0)
            assert np.array_equal(p.asnumpy(), (2 * ref_rng.integers(0, 2, size=n) - 1).astype(dt))
            # the buffered half word is consumed first by the next 32-bit draw -- here on the device again ...
            p2 = ift.from_random(dom, "pm1", dtype=cdt, device_id=0)
            assert np.array_equal(p2.asnumpy(), units[ref_rng.integers(0, 4, size=n)].astype(cdt))
            # ... the complex uniform draws real part first, and a host draw continues exactly where numpy would be
            cu = ift.from_random(dom, "uniform", dtype=cdt, device_id=0)
            ref = ref_rng.uniform(0.0, 1.0, n) + 1j * ref_rng.uniform(0.0, 1.0, n)
            assert np.array_equal(cu.asnumpy(), ref.astype(cdt))
            p3 = ift.from_random(ift.UnstructuredDomain(9), "pm1", dtype=dt, device_id=0)  # small: host draw + upload
            assert np.array_equal(p3.asnumpy(), (2 * ref_rng.integers(0, 2, size=9) - 1).astype(dt))
            assert ift.random.current_rng().bit_generator.state == ref_rng.bit_generator.state
            nrm = ift.from_random(dom, "normal", dtype=dt, device_id=0)
            assert np.allclose(nrm.asnumpy(), ref_rng.normal(0.0, 1.0, n).astype(dt), rtol=1e-6, atol=0)
        finally:
            ift.random.pop_sseq()


# ---- bounded integers: Random.uniform of integer fields (nk_pcg64_integers; reference random.py:252-256) --------------------
INT_RANGES = [(0, 6), (0, 7), (-3, 11), (10, 1000), (0, 2 ** 31), (0, 2 ** 32 - 2), (0, 2 ** 32 - 1), (0, 2 ** 32), (0, 2 ** 40),
              (-2 ** 62, 2 ** 62), (0, 2 ** 63 - 1), (-2 ** 63, 2 ** 63 - 1), (5, 5)]


def _int_emu_draw(state, inc, need, lo, span, nthreads):
    lib = _emu()
    lib.emu_pcg64_integers.restype = ctypes.c_uint
    lib.emu_pcg64_integers.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int64,
                                       ctypes.c_void_p, ctypes.c_void_p]
    words = (ctypes.c_uint64 * 4)(state >> 64, state & (2 ** 64 - 1), inc >> 64, inc & (2 ** 64 - 1))
    out, used = np.zeros(need, dtype=np.int64), ctypes.c_uint64(0)
    err = lib.emu_pcg64_integers(ctypes.addressof(words), ctypes.addressof(words) + 16, need, lo, span, nthreads, out.ctypes.data,
                                 ctypes.addressof(used))
    return out, int(used.value), bool(err & 2)


@pytest.mark.parametrize("low,high", INT_RANGES)
@pytest.mark.parametrize("n,pre", [(1, 0), (45, 1), (5000, 0), (5000, 3)])
def test_bounded_integer_bodies_equal_numpy(low, high, n, pre):
    """The two passes of nk_pcg64_integers (host emulation) + the host bookkeeping around them (backend._bounded_integers:
    PCG64's buffered 32-bit half, retry sizing, the state afterwards) against numpy's Generator.integers, bit for bit --
    values AND generator state: 32-bit Lemire with rejections (ranges near 2^31 reject every other word), the raw-word
    ranges 2^32 - 1 and 2^64 - 1, 64-bit Lemire, the empty range; `pre` bounded draws before set the buffered half."""
    from nifty_amd import backend as B

    a, b = np.random.default_rng(123), np.random.default_rng(123)
    for r in (a, b):
        for _ in range(pre):
            r.integers(0, 4)
    ref = a.integers(low, high, n, endpoint=True)
    first, rest = B._bounded_integers(b, low, high, n, _int_emu_draw)
    got = np.array(list(first) + ([] if rest is None else list(rest)), dtype=np.int64)
    assert np.array_equal(got, ref)
    assert a.bit_generator.state == b.bit_generator.state
    assert a.normal() == b.normal()  # ... and the streams continue together


@pytest.mark.gpu
@pytest.mark.parametrize("low,high", INT_RANGES)
def test_device_bounded_integers_equal_numpy(low, high):
    """backend.pcg64_integers on the GPU against numpy: values, generator state, a following draw."""
    import torch

    from nifty_amd import backend as B

    for n, pre in ((1, 0), (77, 1), (300000, 0), (300001, 1)):
        a, b = np.random.default_rng(9), np.random.default_rng(9)
        for r in (a, b):
            for _ in range(pre):
                r.integers(0, 4)
        ref = a.integers(low, high, n, endpoint=True)
        got = B.pcg64_integers(b, low, high, (n,), torch.device("cuda:0"))
        assert got.dtype == torch.int64 and np.array_equal(got.cpu().numpy(), ref)
        assert a.bit_generator.state == b.bit_generator.state


@pytest.mark.gpu
def test_integer_uniform_fields_on_the_device_equal_the_reference():
    """Field.from_random(..., 'uniform', dtype=int) on a GPU (random.Random.on_device -> nk_pcg64_integers) against what the
    REFERENCE drew for the same seed (tests/golden/small_ops.npz: ranges 7, 8, 15, 991, 2^40 + 1 and the empty range; int64
    and int32) -- small fields on the host path, large ones on the device, both the reference's numbers."""
    import nifty_amd as ift
    from nifty_amd import random as R
    from tests import goldenlib as gl

    z = gl.load("small_ops")
    for tag, dt, shape in (("u7", np.int64, (5, 9)), ("u8", np.int64, (33,)), ("un", np.int64, (4, 4, 4)), ("u32", np.int32, (77,)),
                           ("big", np.int64, (19,)), ("one", np.int64, (6,))):
        low, high = (int(v) for v in z[f"uni.{tag}.args"])
        for min_n in (1 << 15, 1):  # the package's threshold, then every draw on the device
            old, R.DEVICE_DRAW_MIN = R.DEVICE_DRAW_MIN, min_n
            try:
                with ift.random.Context(123):
                    f = ift.Field.from_random(ift.UnstructuredDomain(shape), "uniform", dtype=dt, low=low, high=high, device_id=0)
                    after = ift.random.current_rng().normal(size=3)
            finally:
                R.DEVICE_DRAW_MIN = old
            assert f.device_id == 0 and f.dtype == np.dtype(dt)
            assert np.array_equal(f.asnumpy(), z[f"uni.{tag}"]) and np.array_equal(after, z[f"uni.{tag}.after"])
