"""Device backend: thin Python wrappers over the libniftyk C ABI for torch tensors on a GPU.

This is the counterpart of the reference's native-dispatch seam ``nifty/cl/ducc_dispatch.py``
(hartley / fftn / ifftn / vdot) plus the array-library call sites listed in SURVEY 2.1.  PyTorch is
used for device memory and streams only; every number is produced by a HIP kernel.  There is no
fallback: a CPU tensor handed to these functions raises.
"""
import ctypes
import math
import os

import numpy as np
import torch

from . import _lib as L
from . import config

_DT = {torch.float32: L.NK_F32, torch.float64: L.NK_F64}
_plans = {}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


_current_device = torch.cuda.current_device


def _stream():
    """Handle of torch's current stream on the CURRENT device: every kernel of the library is launched on it, so the
    operands must live on the current device (`_require_device`; `optimize_kl(device_id=k)` and the multi-rank
    initialisation make cuda:k current)."""
    if _raw_stream is not None:  # ~0.3 us instead of ~8 us per launch: the small configs are host-bound
        return _raw_stream(_current_device())
    return torch.cuda.current_stream().cuda_stream


def _wrong_device(index):
    raise RuntimeError(f"nifty_amd.backend: operand on cuda:{index} but the current device is cuda:{_current_device()}; "
                       f"run under `with torch.cuda.device({index})` (kernels launch on the current device's stream)")


def _require_device(*ts):
    cur = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("nifty_amd.backend: device kernels need GPU tensors (no CPU fallback)")
        if cur is None:
            cur = _current_device()
        if t.device.index != cur:
            _wrong_device(t.device.index)


def dtype_code(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype for device kernels: {t.dtype}")


def ptr(t):
    return 0 if t is None else t.data_ptr()


class Plan:
    """An nk_plan plus its workspace (immutable after creation, cached per shape/dtype/device)."""

    def __init__(self, shape, dtype, batch, device):
        lib = L.load()
        self.shape, self.dtype, self.batch, self.device = tuple(shape), dtype, int(batch), device
        self._p = ctypes.c_void_p()
        shp = (ctypes.c_int64 * len(shape))(*shape)
        with torch.cuda.device(device):
            L.check(lib.nk_plan_create(ctypes.byref(self._p), len(shape), shp, _DT[dtype], self.batch), "nk_plan_create")
            self.workspace = torch.empty(lib.nk_plan_workspace_bytes(self._p), dtype=torch.uint8, device=device)

    @property
    def handle(self):
        return self._p

    def __del__(self):
        try:
            if self._p:
                L.load().nk_plan_destroy(self._p)
        except Exception:
            pass


class PlanView:
    """The same nk_plan (twiddle tables, geometry: immutable, shareable) with a workspace of its own, so that transforms of
    the same shape can run on several streams at once (engine.FusedModel lanes)."""

    def __init__(self, plan):
        self._plan = plan
        self.shape, self.dtype, self.batch, self.device = plan.shape, plan.dtype, plan.batch, plan.device
        self.workspace = torch.empty_like(plan.workspace)

    handle = property(lambda self: self._plan.handle)


def get_plan(shape, dtype, batch=1, device=None):
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = (tuple(int(s) for s in shape), dtype, int(batch), device.index)
    p = _plans.get(key)
    if p is None:
        p = _plans[key] = Plan(key[0], dtype, batch, device)
    return p


_unsupported = set()


def plan_supported(shape, dtype, batch=1, device=None):
    """True when the native planner takes this transform (prime factors <= 7, even last axis, line-length limits);
    otherwise the array seam runs the chirp-z composition below and the fused nodes step aside for the generic graph."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = (tuple(int(s) for s in shape), dtype, int(batch), device.index)
    if key in _unsupported:
        return False
    try:
        get_plan(shape, dtype, batch, device)
    except NotImplementedError:
        _unsupported.add(key)
        return False
    return True


# ---- any-length fallback (Bluestein / chirp-z), composed from the native power-of-two c2c kernel -------------------
# The reference's FFT (ducc0) takes every length (test_fft_operator.py:58-103 uses 10, 11, 12).  Lengths the planner
# rejects are rare in the hot path, so they are served axis by axis (element-wise glue: nk_cplx_rows):  X[j] = w[j] * sum_k (x[k] w[k]) conj(w)[j-k]
# with w[k] = exp(-+ i pi k^2 / n) -- a cyclic convolution of power-of-two length m >= 2n-1, i.e. three native c2c
# transforms per axis.  The chirp angles come from k^2 mod 2n in integers, so they carry no large-argument error.
_chirps = {}
_CHIRP_MAX = 1 << 14  # longest padded convolution (power of two) the fallback will build


def _chirp(n, cdt, inverse, device):
    key = (n, cdt, bool(inverse), device.index)
    ent = _chirps.get(key)
    if ent is None:
        m = 1 << (2 * n - 2).bit_length()
        if m > _CHIRP_MAX:
            raise NotImplementedError(f"axis of length {n} is too long for the chirp-z fallback")
        k = torch.arange(n, dtype=torch.int64)
        ang = (k * k % (2 * n)).to(torch.float64) * (math.pi / n)
        w = torch.polar(torch.ones(n, dtype=torch.float64), ang if inverse else -ang)
        b = torch.zeros(m, dtype=torch.complex128)
        b[:n] = w.conj()
        b[m - n + 1:] = w.conj()[1:].flip(0)
        w, b = w.to(cdt).to(device), b.to(cdt).to(device)
        rdt = torch.float32 if cdt == torch.complex64 else torch.float64
        get_plan((m,), rdt, 1, device)  # native power-of-two transform or NotImplementedError -- never the fallback again
        ent = _chirps[key] = (m, w, fftn(b, ndim=1))
    return ent


BLUESTEIN_LDS_BYTES = 64 * 1024  # nk_bluestein_rows keeps one padded row of m complex values in LDS
_blu_tables = {}


def _bluestein_tables(n, m, cdt, inverse, device):
    """(chirp w[n], spectrum of the chirp filter in bit-reversed order [m], twiddles exp(-2 pi i k / m) [m/2]) of
    nk_bluestein_rows -- host set-up in complex128 (angles from k^2 mod 2n in integers), rounded to the field's precision."""
    key = (n, m, cdt, bool(inverse), device.index)
    ent = _blu_tables.get(key)
    if ent is None:
        k = np.arange(n, dtype=np.int64)
        ang = (k * k % (2 * n)).astype(np.float64) * (np.pi / n)
        w = np.exp(1j * ang if inverse else -1j * ang)
        b = np.zeros(m, dtype=np.complex128)
        b[:n] = w.conj()
        b[m - n + 1:] = w.conj()[1:][::-1]
        bhat = np.fft.fft(b)
        bits = m.bit_length() - 1
        rev = np.zeros(m, dtype=np.int64)
        for bit in range(bits):
            rev |= ((np.arange(m) >> bit) & 1) << (bits - 1 - bit)
        tw = np.exp(-2j * np.pi * np.arange(m // 2) / m)
        npdt = np.complex64 if cdt == torch.complex64 else np.complex128
        ent = _blu_tables[key] = tuple(torch.from_numpy(np.ascontiguousarray(a.astype(npdt))).to(device) for a in (w, bhat[rev], tw))
    return ent


def cplx_rows(a, w, in_cols, out_cols, mode, scale=1.0, sgn=1):
    """nk_cplx_rows on a contiguous tensor viewed as rows x in_cols: chirp multiply with zero padding / cropping
    (mode 0), real -> complex promotion (mode 1), Hartley combine Re + sgn Im (mode 2)."""
    _require_device(a, w)
    rows = a.numel() // max(1, in_cols)
    if mode == 2:
        rdt = torch.float32 if a.dtype == torch.complex64 else torch.float64
        out = torch.empty(a.shape[:-1] + (out_cols,), dtype=rdt, device=a.device)
    else:
        cdt = a.dtype if a.is_complex() else (torch.complex64 if a.dtype == torch.float32 else torch.complex128)
        out = torch.empty(a.shape[:-1] + (out_cols,), dtype=cdt, device=a.device)
    code = 0 if out.dtype in (torch.float32, torch.complex64) else 1
    L.check(L.load().nk_cplx_rows(rows, in_cols, out_cols, a.data_ptr(), ptr(w), out.data_ptr(), mode, float(scale),
                                  int(sgn), code, _stream()), "nk_cplx_rows")
    return out


def _fft_last_axis_any(z, inverse, real_in=False, hartley=0, scale=1.0):
    """Unnormalised c2c transform over the last axis of a contiguous tensor, any length.  real_in: z is real (the first axis
    of a real transform); hartley = +-1: return the real Hartley combination Re + hartley Im times `scale` (the last axis).
    Axes served by nk_bluestein_rows take these ends inside the launch, the others through nk_cplx_rows."""
    n = z.shape[-1]
    rdt = z.dtype if real_in else (torch.float32 if z.dtype == torch.complex64 else torch.float64)
    cdt = torch.complex64 if rdt == torch.float32 else torch.complex128
    batch = z.numel() // max(1, n)
    m = 1 << max(2, (2 * n - 2).bit_length())
    native = n == 1 or z.numel() == 0 or plan_supported((n,), rdt, batch, z.device)
    if (not native and m * (8 if rdt == torch.float32 else 16) <= BLUESTEIN_LDS_BYTES and os.environ.get("NK_BLUESTEIN", "1") != "0"):
        # ONE launch: the padded rows stay in LDS through both transforms of the convolution (nk_bluestein_rows)
        w, bhat_br, tw = _bluestein_tables(n, m, cdt, inverse, z.device)
        out = torch.empty(z.shape, dtype=rdt if hartley else cdt, device=z.device)
        L.check(L.load().nk_bluestein_rows(batch, n, m, z.data_ptr(), w.data_ptr(), bhat_br.data_ptr(), tw.data_ptr(), out.data_ptr(),
                                           float(scale) if hartley else 1.0, 1 if real_in else 0, int(hartley),
                                           0 if rdt == torch.float32 else 1, _stream()), "nk_bluestein_rows")
        return out
    if real_in:
        z = cplx_rows(z, None, n, n, 1)
    if n == 1 or z.numel() == 0:
        res = z.clone()
    elif native:
        res = fftn(z, ndim=1, inverse=inverse)
    else:
        # check BEFORE anything of length m is built: an unsupported m must end here, not re-enter this fallback
        if m > _CHIRP_MAX or not plan_supported((m,), rdt, batch, z.device) or not plan_supported((m,), rdt, 1, z.device):
            raise NotImplementedError(f"axis of length {n} is too long for the chirp-z fallback")
        m, w, fb = _chirp(n, z.dtype, inverse, z.device)
        a = cplx_rows(z, w, n, m, 0)                       # x[k] w[k], zero-padded to m
        p = cplx_rows(fftn(a, ndim=1), fb, m, m, 0)        # times the transformed chirp
        c = fftn(p, ndim=1, inverse=True, scale=1.0 / m)   # cyclic convolution
        res = cplx_rows(c, w, m, n, 0)                     # first n entries, times w[j]
    return cplx_rows(res, None, n, n, 2, scale, hartley) if hartley else res


def _fft_any(z, ndim, inverse, real_in=False, hartley=0, scale=1.0):
    first, last = z.dim() - ndim, z.dim() - 1
    for ax in range(first, z.dim()):
        z = _fft_last_axis_any(z.movedim(ax, -1).contiguous(), inverse, real_in=real_in and ax == first,
                               hartley=hartley if ax == last else 0, scale=scale).movedim(-1, ax)
    return z.contiguous()


def _convention():
    return 0 if config.get("hartley_convention") == "non_canonical_hartley" else 1


def hartley(x, ndim=None, scale=1.0, out=None):
    """Genuine N-D Hartley transform over the last ``ndim`` axes (reference ducc_dispatch.hartley)."""
    _require_device(x)
    x = x.contiguous()
    ndim = x.dim() if ndim is None else ndim
    shape = x.shape[x.dim() - ndim:]
    batch = x.numel() // max(1, int(torch.Size(shape).numel()))
    if not plan_supported(shape, x.dtype, batch, x.device):
        h = _fft_any(x, ndim, False, real_in=True, hartley=1 if _convention() == 0 else -1, scale=scale)
        return h if out is None else out.copy_(h)
    plan = get_plan(shape, x.dtype, batch, x.device)
    out = torch.empty_like(x) if out is None else out
    L.check(L.load().nk_hartley(plan.handle, x.data_ptr(), out.data_ptr(), float(scale), _convention(),
                                plan.workspace.data_ptr(), _stream()), "nk_hartley")
    return out


def hartley_fused(plan, fuse):
    if plan.device.index != _current_device():
        _wrong_device(plan.device.index)
    L.check(L.load().nk_hartley_fused(plan.handle, ctypes.byref(fuse), _convention(), plan.workspace.data_ptr(),
                                      _stream()), "nk_hartley_fused")


def plan_sandwich(plan):
    """True when nk_hartley_sandwich serves this plan (>= 2 axes, power-of-two lengths 64 .. 4096)."""
    return bool(L.load().nk_plan_sandwich(plan.handle))


def hartley_sandwich(plan, fuse, scale_first):
    """EPI(fuse.scale * H(mul_scalar * mul . scale_first * H(PRO(in)))) in five passes (include/niftyk.h)."""
    if plan.device.index != _current_device():
        _wrong_device(plan.device.index)
    L.check(L.load().nk_hartley_sandwich(plan.handle, ctypes.byref(fuse), float(scale_first), _convention(),
                                         plan.workspace.data_ptr(), _stream()), "nk_hartley_sandwich")


def hartley_sandwich_pair(plan, fuse_a, fuse_b, scale_first, workspace_b):
    """Two sandwiches accumulating into one output with their final passes in ONE launch (nk_hartley_sandwich_pair): same
    bits as two hartley_sandwich calls; `workspace_b` is a second workspace of the plan's size for sample B."""
    if plan.device.index != _current_device():
        _wrong_device(plan.device.index)
    L.check(L.load().nk_hartley_sandwich_pair(plan.handle, ctypes.byref(fuse_a), ctypes.byref(fuse_b), float(scale_first),
                                              _convention(), plan.workspace.data_ptr(), workspace_b.data_ptr(), _stream()),
            "nk_hartley_sandwich_pair")


def fftn(x, ndim=None, inverse=False, scale=1.0):
    """c2c FFT over the last ``ndim`` axes of a complex tensor (reference ducc_dispatch.fftn / ifftn)."""
    _require_device(x)
    if not x.is_complex():
        raise TypeError("fftn expects a complex tensor")
    x = x.contiguous()
    ndim = x.dim() if ndim is None else ndim
    shape = x.shape[x.dim() - ndim:]
    batch = x.numel() // max(1, int(torch.Size(shape).numel()))
    rdt = torch.float32 if x.dtype == torch.complex64 else torch.float64
    if not plan_supported(shape, rdt, batch, x.device):
        f = _fft_any(x, ndim, inverse)
        return f if float(scale) == 1.0 else cplx_rows(f, None, f.shape[-1], f.shape[-1], 0, scale)
    plan = get_plan(shape, rdt, batch, x.device)
    out = torch.empty_like(x)
    L.check(L.load().nk_fftn(plan.handle, torch.view_as_real(x).data_ptr(), torch.view_as_real(out).data_ptr(),
                             1 if inverse else 0, float(scale), plan.workspace.data_ptr(), _stream()), "nk_fftn")
    return out


def new_scalar(device, n=1):
    return torch.zeros(n, dtype=torch.float64, device=device)


def vdot(a, b, result=None, accumulate=False):
    """sum(a*b) with fp64 accumulation into a DEVICE double (no host sync)."""
    _require_device(a, b)
    if a.numel() != b.numel() or a.dtype != b.dtype:
        raise ValueError("vdot: shape/dtype mismatch")
    result = new_scalar(a.device) if result is None else result
    L.check(L.load().nk_vdot(a.numel(), a.data_ptr(), b.data_ptr(), dtype_code(a), result.data_ptr(),
                             1 if accumulate else 0, _stream()), "nk_vdot")
    return result


def vsum(a, result=None, accumulate=False):
    _require_device(a)
    result = new_scalar(a.device) if result is None else result
    L.check(L.load().nk_sum(a.numel(), a.data_ptr(), dtype_code(a), result.data_ptr(), 1 if accumulate else 0,
                            _stream()), "nk_sum")
    return result


def binary(op, a, b, out=None):
    """a (op) b for tensors of identical shape/dtype, or tensor (op) python scalar."""
    ta, tb = torch.is_tensor(a), torch.is_tensor(b)
    ref = a if ta else b
    _require_device(ref)
    if ta and tb and (a.shape != b.shape or a.dtype != b.dtype):
        raise ValueError("binary: shape/dtype mismatch")
    out = torch.empty_like(ref) if out is None else out
    L.check(L.load().nk_binary(op, ref.numel(), ptr(a) if ta else 0, 0.0 if ta else float(a), ptr(b) if tb else 0,
                               0.0 if tb else float(b), out.data_ptr(), dtype_code(ref), _stream()), "nk_binary")
    return out


def _cplx_operand(x, cdt):
    """(pointer, kind, re, im) of an operand of cplx_muldiv: complex tensor, real tensor or python scalar"""
    if torch.is_tensor(x):
        if x.is_complex():
            return x.to(cdt).contiguous(), 0, 0.0, 0.0
        rdt = torch.float32 if cdt == torch.complex64 else torch.float64
        return x.to(rdt).contiguous(), 1, 0.0, 0.0
    z = complex(x)
    return None, 2, z.real, z.imag


def cplx_muldiv(a, b, divide=False, conj_b=False):
    """a * b or a / b (b optionally conjugated) as a complex tensor; a, b: complex tensors, real tensors of the same shape or
    python scalars, at least one of them complex (nk_cplx_muldiv)."""
    tensors = [t for t in (a, b) if torch.is_tensor(t)]
    _require_device(*tensors)
    wide = any(t.dtype in (torch.complex128, torch.float64) for t in tensors)
    cdt = torch.complex128 if wide else torch.complex64
    ta, ka, ar, ai = _cplx_operand(a, cdt)
    tb, kb, br, bi = _cplx_operand(b, cdt)
    shape = tensors[0].shape
    if any(t.shape != shape for t in tensors):
        raise ValueError("cplx_muldiv: shape mismatch")
    out = torch.empty(shape, dtype=cdt, device=tensors[0].device)
    L.check(L.load().nk_cplx_muldiv(out.numel(), ptr(ta), ka, ar, ai, ptr(tb), kb, br, bi, int(bool(conj_b)), int(bool(divide)),
                                    out.data_ptr(), 0 if cdt == torch.complex64 else 1, _stream()), "nk_cplx_muldiv")
    return out


CPLX_POINTWISE = {"exp": 0, "log": 1, "sqrt": 2, "reciprocal": 3, "conjugate": 4, "abs": 5, "absolute": 5}


def cplx_pointwise(op, x):
    """exp / log / sqrt / reciprocal / conjugate of a complex tensor, or its modulus as a REAL tensor (nk_cplx_pointwise)."""
    _require_device(x)
    x = x.contiguous()
    fn = CPLX_POINTWISE[op]
    rdt = torch.float32 if x.dtype == torch.complex64 else torch.float64
    out = torch.empty(x.shape, dtype=rdt if fn == 5 else x.dtype, device=x.device)
    L.check(L.load().nk_cplx_pointwise(fn, x.numel(), x.data_ptr(), out.data_ptr(), 0 if x.dtype == torch.complex64 else 1,
                                       _stream()), "nk_cplx_pointwise")
    return out


def axpby(alpha, x, beta=0.0, y=None, out=None):
    _require_device(x, y)
    out = torch.empty_like(x) if out is None else out
    L.check(L.load().nk_axpby(x.numel(), float(alpha), x.data_ptr(), float(beta), ptr(y), out.data_ptr(), dtype_code(x),
                              _stream()), "nk_axpby")
    return out


def axpby_sqnorm(alpha, x, beta, y, result, accumulate=False):
    """alpha x + beta y (a new tensor) and result (+)= |that|^2 in the same pass (device double)."""
    _require_device(x, y, result)
    out = torch.empty_like(x)
    L.check(L.load().nk_axpby_sqnorm(x.numel(), float(alpha), x.data_ptr(), float(beta), y.data_ptr(), out.data_ptr(),
                                     dtype_code(x), result.data_ptr(), 1 if accumulate else 0, _stream()), "nk_axpby_sqnorm")
    return out


POINTWISE = {"exp": 0, "log": 1, "sqrt": 2, "tanh": 3, "sigmoid": 4, "reciprocal": 5, "power": 6, "abs": 7,
             "absolute": 7, "log1p": 8, "expm1": 9, "arctan": 10, "sin": 11, "cos": 12, "tan": 13, "sinc": 14, "log10": 15,
             "sinh": 16, "cosh": 17, "sign": 18, "softplus": 19, "exponentiate": 20, "unitstep": 21, "clip": 22}


def pointwise(name, x, param=0.0, want_derivative=False, param2=None):
    """Pointwise function (and derivative) of a real device tensor; `clip` takes its bounds in (param, param2), None = open"""
    _require_device(x)
    fx = torch.empty_like(x)
    dfx = torch.empty_like(x) if want_derivative else None
    if name == "clip":
        lo = -math.inf if param is None else float(param)
        hi = math.inf if param2 is None else float(param2)
        L.check(L.load().nk_clip(lo, hi, x.numel(), x.data_ptr(), fx.data_ptr(), ptr(dfx), dtype_code(x), _stream()), "nk_clip")
    else:
        L.check(L.load().nk_pointwise(POINTWISE[name], float(param), x.numel(), x.data_ptr(), fx.data_ptr(), ptr(dfx),
                                      dtype_code(x), _stream()), "nk_pointwise")
    return (fx, dfx) if want_derivative else fx


def gather(table, pidx, out_shape, out=None):
    _require_device(table, pidx)
    out = torch.empty(out_shape, dtype=table.dtype, device=table.device) if out is None else out
    L.check(L.load().nk_gather(out.numel(), table.data_ptr(), pidx.data_ptr(), out.data_ptr(), dtype_code(table),
                               _stream()), "nk_gather")
    return out


def scatter_add(x, pidx, nbins):
    """np.bincount(pidx, weights=x, minlength=nbins): fp64 bins, by fp64 atomics.  Exact and reproducible when the indices
    are unique (MaskOperator); with collisions the last bit depends on the order -- static index maps go through
    ``bin_plan`` / ``bin_sum`` instead."""
    _require_device(x, pidx)
    bins = torch.zeros(nbins, dtype=torch.float64, device=x.device)
    L.check(L.load().nk_scatter_add(x.numel(), x.data_ptr(), pidx.data_ptr(), nbins, bins.data_ptr(), dtype_code(x),
                                    _stream()), "nk_scatter_add")
    return bins


def lanes_for(nnz, nrows):
    """Lanes per row of nk_csr_rowsum from the average row length."""
    avg = nnz / max(1, nrows)
    return 1 if avg <= 4 else 4 if avg <= 32 else 16 if avg <= 256 else 64


BIN_PLAN_MAX = 1 << 28  # largest static index map that is sorted once (the sort holds ~20 bytes per point transiently)


def bin_plan(pidx, nbins):
    """(rowptr int64[nbins+1], perm int32[n], lanes) of a STATIC device index map: the source points listed bin by bin
    (stable sort, made once), so that ``bin_sum`` adds every bin in a fixed order without atomics (the scatter-add of
    np.bincount / _special_add_at, reference utilities.py:222-246, turned inside out).  Index bookkeeping only (set-up)."""
    _require_device(pidx)
    flat = pidx.reshape(-1)
    perm = torch.argsort(flat, stable=True).to(torch.int32)
    rowptr = torch.zeros(nbins + 1, dtype=torch.int64, device=pidx.device)
    rowptr[1:] = torch.cumsum(torch.bincount(flat, minlength=nbins), 0)
    return rowptr, perm, lanes_for(flat.numel(), nbins)


def bin_sum(x, plan, out_dtype=None):
    """bins[b] = sum of x over the points of bin b (plan from ``bin_plan``), fp64 accumulation in a fixed order; returned in
    x's dtype (the reference casts its fp64 bincount back to the field dtype as well, distributors.py:112)."""
    rowptr, perm, lanes = plan
    _require_device(x, perm)
    y = torch.empty(rowptr.numel() - 1, dtype=x.dtype, device=x.device)
    L.check(L.load().nk_csr_rowsum(y.numel(), rowptr.data_ptr(), perm.data_ptr(), 0, x.data_ptr(), y.data_ptr(),
                                   dtype_code(x), lanes, _stream()), "nk_csr_rowsum")
    return y


def spmv(rowptr, col, wgt, x, nrows, lanes=64):
    """y = R x for a CSR response (LOSResponse TIMES; ADJOINT_TIMES with the transposed arrays): fixed summation order."""
    _require_device(x, col)
    y = torch.empty(nrows, dtype=x.dtype, device=x.device)
    L.check(L.load().nk_csr_rowsum(nrows, rowptr.data_ptr(), col.data_ptr(), wgt.data_ptr(), x.data_ptr(), y.data_ptr(),
                                   dtype_code(x), lanes, _stream()), "nk_csr_rowsum")
    return y


class TiledMatrix:
    """Device copy of a ``los_response.tiled_plan`` (the arrays of ``nk_tiled_csr``) + the partial-sum scratch of its launches
    (one set per member of a batched call, grown on demand; launches on ONE stream at a time, like every workspace here)."""

    _keys = ("item_tile", "item_blk", "blk_slot", "row_slot", "loc", "wgt")

    def __init__(self, plan, device):
        self.n_rows, self.n_slots = int(plan["n_rows"]), int(plan["n_slots"])
        self.nnz, self.n_padded = int(plan["nnz"]), int(len(plan["loc"]))
        self._arrays = {}
        for k in self._keys:
            a = plan[k]
            if a.dtype == np.uint16:  # (torch has no uint16 arithmetic; the bytes are what the kernel reads)
                a = a.view(np.int16)
            self._arrays[k] = torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.c = L.TiledCsr(n_rows=self.n_rows, n_slots=self.n_slots, n_items=int(plan["n_items"]), ny=int(plan["ny"]),
                            nx=int(plan["nx"]), th=int(plan["th"]), tw=int(plan["tw"]),
                            **{k: ptr(self._arrays[k]) for k in self._keys})
        self._scratch = {}

    def _scratch_for(self, count, device):
        key = (str(device), torch.cuda.current_stream(device).cuda_stream)
        buf = self._scratch.get(key)
        if buf is None or buf.numel() < count * max(self.n_slots, 1):
            buf = self._scratch[key] = torch.empty(count * max(self.n_slots, 1), dtype=torch.float64, device=device)
        return buf

    def rowsum(self, xs, ys):
        """ys[m] = R xs[m] for up to MAX_BATCH members sharing the matrix."""
        _require_device(*xs)
        scratch = self._scratch_for(len(xs), xs[0].device)
        L.check(L.load().nk_tiled_rowsum(ctypes.byref(self.c), len(xs), L.ptr_array(xs), L.ptr_array(ys), scratch.data_ptr(),
                                         dtype_code(xs[0]), _stream()), "nk_tiled_rowsum")
        return ys


def spmv_t(rowptr, col, wgt, y, ncols):
    """x = R^T y by fp64 atomics, returned in y's dtype -- for callers without the transposed arrays (LOSResponse holds
    them and uses ``spmv``); order-dependent in the last bit."""
    _require_device(y, col)
    x64 = torch.zeros(ncols, dtype=torch.float64, device=y.device)
    L.check(L.load().nk_spmv_t(y.numel(), rowptr.data_ptr(), col.data_ptr(), wgt.data_ptr(), y.data_ptr(), x64.data_ptr(),
                               dtype_code(y), _stream()), "nk_spmv_t")
    return x64.to(y.dtype)  # dtype conversion copy only (same convention as scatter_add's callers)


def roll(x, shifts):
    """numpy.roll(x, shifts, axes=all) for a device tensor of any element type of 4, 8 or 16 bytes (real or complex): one
    nk_roll launch, a fresh array."""
    _require_device(x)
    x = x.contiguous()
    if x.dim() == 0 or x.dim() > 6 or x.element_size() not in (4, 8, 16):
        raise NotImplementedError("roll: 1 to 6 axes, elements of 4, 8 or 16 bytes")
    out = torch.empty_like(x)
    nd = x.dim()
    L.check(L.load().nk_roll(nd, (ctypes.c_int64 * nd)(*x.shape), (ctypes.c_int64 * nd)(*[int(v) for v in shifts]), x.element_size(),
                             x.data_ptr(), out.data_ptr(), _stream()), "nk_roll")
    return out


def stats(x):
    """(sum, sum of squares, number of ignored entries) over the entries of x that are neither NaN nor 0."""
    _require_device(x)
    res = torch.empty(3, dtype=torch.float64, device=x.device)
    xc = x.contiguous()
    L.check(L.load().nk_stats(xc.numel(), xc.data_ptr(), dtype_code(xc), res.data_ptr(), _stream()), "nk_stats")
    s, s2, nign = res.cpu().tolist()
    return s, s2, int(round(nign))


def cumsum(x, reverse=False, out=None):
    """Inclusive prefix (reverse: suffix) sums of a contiguous 1-D device tensor."""
    _require_device(x)
    x = x.contiguous()
    out = torch.empty_like(x) if out is None else out
    L.check(L.load().nk_cumsum(x.numel(), x.data_ptr(), out.data_ptr(), 1 if reverse else 0, dtype_code(x), _stream()),
            "nk_cumsum")
    return out


_PCG_MULT = 0x2360ED051FC65DA44385DF649FCCF645
_M128 = (1 << 128) - 1


def _pcg64_advanced(state, inc, delta):
    """State of a PCG64 stream `delta` steps later (LCG jump-ahead in O(log delta) steps)."""
    am, ap, m, p = 1, 0, _PCG_MULT, inc
    while delta:
        if delta & 1:
            am = (am * m) & _M128
            ap = (ap * m + p) & _M128
        p = ((m + 1) * p) & _M128
        m = (m * m) & _M128
        delta >>= 1
    return (am * state + ap) & _M128


def _pcg64_output(state):
    """PCG64's output function (XSL-RR 128/64) of a state."""
    hi, lo = state >> 64, state & (2**64 - 1)
    x, rot = hi ^ lo, hi >> 58
    return ((x >> rot) | (x << ((64 - rot) & 63))) & (2**64 - 1)


def _bounded_integers(rng, low, high, n, draw):
    """numpy's `rng.integers(low, high + 1, n)` (int64; the reference's Random.uniform of integer fields, random.py:252-256)
    with the words of the stream supplied by `draw(state, inc, n, low, span, nthreads) -> (values, words_used, short)`:
    the device kernels (pcg64_integers) or their host emulation (tests).  Handles what is not stream arithmetic: the empty
    range, PCG64's buffered 32-bit half (next_uint32 returns the cached high half of the previous draw first), the sizing /
    retry of the walk for rejected words, and the generator state afterwards -- `rng` is left exactly as numpy leaves it.
    Returns (first, rest): `first` = values produced on the host from the cached half-word (0 or 1), `rest` from `draw`."""
    bg = rng.bit_generator
    st = bg.state
    if st.get("bit_generator") != "PCG64":
        raise TypeError("needs a numpy Generator over PCG64 (np.random.default_rng)")
    low, high = int(low), int(high)
    if high < low:
        raise ValueError("low > high")
    span = high - low
    if span >= 2**64:
        raise ValueError("range too large")
    if n == 0 or span == 0:
        return [low] * (0 if span else n), None
    state, inc = int(st["state"]["state"]), int(st["state"]["inc"])
    wide = span > 0xFFFFFFFF
    first = []
    if not wide and st["has_uint32"]:
        # the cached half-word is the first 32-bit word of this call
        st["has_uint32"], word = 0, int(st["uinteger"])
        if span == 0xFFFFFFFF:
            first.append(low + word)
        else:
            m = word * (span + 1)
            if (m & 0xFFFFFFFF) >= (0xFFFFFFFF - span) % (span + 1):
                first.append(low + (m >> 32))
    need = n - len(first)
    rest, words = None, 0
    if need > 0:
        full = 2**64 if wide else 2**32
        p_rej = 0.0 if span + 1 == full else ((full - 1 - span) % (span + 1)) / full
        per = 32 if wide else 64  # words per thread (NK_RNG_FIX raw draws)
        for attempt in range(12):
            expect = need / (1.0 - p_rej)
            nwords = int(expect + (2 ** attempt) * (8.0 * (expect * p_rej) ** 0.5 + 64))
            rest, words, short = draw(state, inc, need, low, span, -(-nwords // per))
            if not short:
                break
        else:
            raise RuntimeError("bounded integers: the walk stayed short of the requested count")
    raws = words if wide else -(-words // 2)
    st["state"]["state"] = _pcg64_advanced(state, inc, raws)
    if not wide and raws > 0:
        # next_uint32 parks the high half of every draw it takes; after an odd number of words that half is still pending
        # (even: the field keeps the stale value, like numpy's state dict does)
        st["has_uint32"], st["uinteger"] = words % 2, _pcg64_output(st["state"]["state"]) >> 32
    bg.state = st
    return first, rest


def pcg64_integers(rng, low, high, shape, device):
    """`rng.integers(low, high + 1, shape)` (int64) of a numpy Generator over PCG64 on `device`: the same values, the same
    generator state afterwards (nk_pcg64_integers; one host sync for the number of words consumed)."""
    n = int(np.prod(shape, dtype=np.int64)) if len(tuple(shape)) else 1
    out = torch.empty(n, dtype=torch.int64, device=device)
    _require_device(out)
    lib = L.load()
    status = torch.empty(2, dtype=torch.int64, device=device)

    def draw(state, inc, need, lo, span, nthreads):
        words = (ctypes.c_uint64 * 4)(state >> 64, state & (2**64 - 1), inc >> 64, inc & (2**64 - 1))
        scratch = torch.empty(lib.nk_pcg64_integers_scratch_bytes(nthreads), dtype=torch.uint8, device=device)
        dst = out[n - need:]
        L.check(lib.nk_pcg64_integers(ctypes.addressof(words), ctypes.addressof(words) + 16, need, lo, span, nthreads, dst.data_ptr(),
                                      scratch.data_ptr(), status.data_ptr(), _stream()), "nk_pcg64_integers")
        used, err = status.cpu().tolist()
        return dst, int(used), bool(err & 2)

    first, _ = _bounded_integers(rng, low, high, n, draw)
    if n and int(high) == int(low):
        out.fill_(int(low))
    elif first:
        out[:len(first)] = torch.tensor(first, dtype=torch.int64)
    return out.reshape(tuple(shape))


def pcg64_normal(rng, mean, std, shape, dtype, device):
    """`rng.normal(mean, std, shape)` of a numpy Generator over PCG64 (the reference's generator: nifty/cl/random.py:146-206
    push_sseq -> np.random.default_rng), computed on `device` from the generator's current state: the same values draw for
    draw (cast to dtype like `.astype`; bit-identical except in the ziggurat tail |x| > 3.654, 2.7e-4 of the draws, where
    the device log1p may differ from the host libm by <= 4 ulp), and `rng` is left in the state the host call would leave it
    in.  std < 0 raises ValueError like numpy does.  nk_pcg64_normal;
    one host sync (the number of raw draws consumed comes back from the device)."""
    import ctypes

    import numpy as np

    bg = rng.bit_generator
    st = bg.state
    if st.get("bit_generator") != "PCG64":
        raise TypeError("pcg64_normal needs a numpy Generator over PCG64 (np.random.default_rng)")
    if float(std) < 0:
        raise ValueError("scale < 0")
    n = int(np.prod(shape, dtype=np.int64)) if len(tuple(shape)) else 1
    out = torch.empty(tuple(shape), dtype=dtype, device=device)
    _require_device(out)
    if n == 0:
        return out
    state, inc = int(st["state"]["state"]), int(st["state"]["inc"])
    words = (ctypes.c_uint64 * 4)(state >> 64, state & (2**64 - 1), inc >> 64, inc & (2**64 - 1))
    status = torch.empty(2, dtype=torch.int64, device=device)
    lib = L.load()
    for attempt in range(9):
        nbytes = lib.nk_pcg64_normal_scratch_bytes(n, attempt)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=device)
        L.check(lib.nk_pcg64_normal(ctypes.addressof(words), ctypes.addressof(words) + 16, n, float(mean), float(std),
                                    out.data_ptr(), dtype_code(out), scratch.data_ptr(), nbytes, attempt,
                                    status.data_ptr(), _stream()), "nk_pcg64_normal")
        consumed, err = status.cpu().tolist()
        del scratch
        if err == 0:
            break
        if err & 1:
            raise RuntimeError("nk_pcg64_normal: chain merge failed")
    else:
        raise RuntimeError("nk_pcg64_normal: scratch sizing failed")
    st["state"]["state"] = _pcg64_advanced(state, inc, int(consumed))
    bg.state = st  # keeps has_uint32 / uinteger, exactly like the host draw of doubles does
    return out


def _pcg64_words(rng):
    bg = rng.bit_generator
    st = bg.state
    if st.get("bit_generator") != "PCG64":
        raise TypeError("needs a numpy Generator over PCG64 (np.random.default_rng)")
    import ctypes

    state, inc = int(st["state"]["state"]), int(st["state"]["inc"])
    words = (ctypes.c_uint64 * 4)(state >> 64, state & (2**64 - 1), inc >> 64, inc & (2**64 - 1))
    return bg, st, state, inc, words


def pcg64_uniform(rng, low, high, shape, dtype, device):
    """`rng.uniform(low, high, shape)` of a numpy Generator over PCG64 (the reference's Random.uniform for real fields,
    nifty/cl/random.py:249-258) computed on `device`: bit-identical values (one raw draw each, low + (high - low) * u without
    contraction, cast like `.astype`), generator advanced by exactly n draws.  nk_pcg64_uniform; no host sync."""
    import ctypes

    import numpy as np

    bg, st, state, inc, words = _pcg64_words(rng)
    n = int(np.prod(shape, dtype=np.int64)) if len(tuple(shape)) else 1
    out = torch.empty(tuple(shape), dtype=dtype, device=device)
    _require_device(out)
    if n == 0:
        return out
    lib = L.load()
    scratch = torch.empty(lib.nk_pcg64_fixed_scratch_bytes(), dtype=torch.uint8, device=device)
    L.check(lib.nk_pcg64_uniform(ctypes.addressof(words), ctypes.addressof(words) + 16, n, float(low), float(high),
                                 out.data_ptr(), dtype_code(out), scratch.data_ptr(), _stream()), "nk_pcg64_uniform")
    st["state"]["state"] = _pcg64_advanced(state, inc, n)
    bg.state = st
    return out


def pcg64_pm1(rng, shape, dtype, device, complex_units=False):
    """The reference's Random.pm1 (nifty/cl/random.py:239-247) on `device`: `2 * rng.integers(0, 2, shape) - 1`, or for
    complex fields one of 1, i, -1, -i from `rng.integers(0, 4, shape)` -- numpy's buffered 32-bit bounded draws (two outputs
    per raw 64-bit value, low half first), bit-identical, with the generator's raw position AND its buffered half word
    (`has_uint32` / `uinteger`) left as the host call would leave them.  `dtype`: the real torch dtype (float32 / float64);
    complex_units returns a complex tensor."""
    import ctypes

    import numpy as np

    n = int(np.prod(shape, dtype=np.int64)) if len(tuple(shape)) else 1
    cdt = {torch.float32: torch.complex64, torch.float64: torch.complex128}[dtype]
    out = torch.empty(tuple(shape), dtype=cdt if complex_units else dtype, device=device)
    _require_device(out)
    if n == 0:
        return out
    flat = (torch.view_as_real(out) if complex_units else out).reshape(-1)
    width = 2 if complex_units else 1
    done = 0
    if rng.bit_generator.state["has_uint32"]:
        # the first output comes from the half word numpy has buffered: let numpy draw it (one value, on the host)
        v = int(rng.integers(0, 4 if complex_units else 2))
        first = ([1, 0], [0, 1], [-1, 0], [0, -1])[v] if complex_units else [2 * v - 1]
        flat[:width] = torch.tensor(first, dtype=dtype, device=device)
        done = 1
    m = n - done
    if m > 0:
        bg, st, state, inc, words = _pcg64_words(rng)
        lib = L.load()
        scratch = torch.empty(lib.nk_pcg64_fixed_scratch_bytes(), dtype=torch.uint8, device=device)
        L.check(lib.nk_pcg64_pm1(ctypes.addressof(words), ctypes.addressof(words) + 16, m, flat[done * width:].data_ptr(),
                                 _DT[dtype], 1 if complex_units else 0, scratch.data_ptr(), _stream()), "nk_pcg64_pm1")
        nraw = (m + 1) // 2
        if m % 2:
            # the high half of the last raw value stays buffered for the next 32-bit draw
            st["state"]["state"] = _pcg64_advanced(state, inc, nraw - 1)
            bg.state = st
            last = int(bg.random_raw(1)[0])  # advances to nraw
            st = bg.state
            st["has_uint32"], st["uinteger"] = 1, last >> 32
            bg.state = st
        else:
            st["state"]["state"] = _pcg64_advanced(state, inc, nraw)
            st["has_uint32"], st["uinteger"] = 0, 0
            bg.state = st
    return out


# ---- operands on another GPU than torch's current one ---------------------------------------------------------------
# The reference lets Fields live on any device_id; kernels here launch on the CURRENT device's stream.  Instead of refusing
# such operands every public entry point makes the device of its first tensor argument current for the
# duration of the call -- a no-op (one attribute read per call) in the common case that it already is.
def _runs_on_operand_device(fn):
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kw):
        for a in args:
            if torch.is_tensor(a):
                if a.is_cuda and a.device.index != _current_device():
                    with torch.cuda.device(a.device):
                        return fn(*args, **kw)
                break
            if isinstance(a, (Plan, PlanView)):
                if a.device.index != _current_device():
                    with torch.cuda.device(a.device):
                        return fn(*args, **kw)
                break
        return fn(*args, **kw)

    return wrapper


for _name in ("cplx_rows", "cplx_pointwise", "hartley", "hartley_fused", "hartley_sandwich", "hartley_sandwich_pair", "fftn", "vdot", "vsum", "binary", "axpby",
              "axpby_sqnorm",
              "pointwise", "gather", "scatter_add", "bin_plan", "bin_sum", "spmv", "spmv_t", "stats", "cumsum"):
    globals()[_name] = _runs_on_operand_device(globals()[_name])
del _name
