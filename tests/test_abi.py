"""The C-ABI library loads and exports every symbol include/niftyk.h declares (no compute calls: CPU box)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "niftyk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nk_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from nifty_amd import _lib

    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 20
    for name in syms:
        assert hasattr(lib, name), f"{name} declared in include/niftyk.h but not exported"
    # the ctypes binding covers the header exactly
    assert sorted(_lib.SIGNATURES) == syms
    loaded = _lib.load()
    assert loaded.nk_version() >= 100


def test_fuse_struct_layout_matches_header():
    from nifty_amd import _lib

    text = open(os.path.join(ROOT, "include", "niftyk.h")).read()
    body = text[text.index("typedef struct nk_fuse {"):text.index("} nk_fuse;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for line in body.splitlines()[1:]:
        line = line.strip().rstrip(";")
        if not line:
            continue
        decl = line.split("(")[0]
        for part in decl.replace("*", " ").split(",") if "," in decl else [decl]:
            names.append(part.replace("*", " ").split()[-1])
    mine = [n.rstrip("_") for n, _ in _lib.Fuse._fields_]
    assert mine == names


def test_plan_errors_without_gpu_are_clean():
    """Argument validation happens before any device call, so it can be exercised on a CPU-only box."""
    from nifty_amd import _lib

    lib = _lib.load()
    p = ctypes.c_void_p()
    shp = (ctypes.c_int64 * 1)(22)  # prime factor 11: only 2, 3, 5, 7 are implemented
    rc = lib.nk_plan_create(ctypes.byref(p), 1, shp, _lib.NK_F64, 1)
    assert rc == _lib.NK_ERR_UNSUPPORTED and b"factor into 2, 3, 5 and 7" in lib.nk_last_error()
    shp = (ctypes.c_int64 * 1)(15)  # odd last axis
    rc = lib.nk_plan_create(ctypes.byref(p), 1, shp, _lib.NK_F64, 1)
    assert rc == _lib.NK_ERR_UNSUPPORTED and b"even length" in lib.nk_last_error()
    rc = lib.nk_plan_create(ctypes.byref(p), 4, shp, _lib.NK_F64, 1)
    assert rc == _lib.NK_ERR_UNSUPPORTED
    rc = lib.nk_vdot(-1, None, None, _lib.NK_F64, None, 0, None)
    assert rc == _lib.NK_ERR_INVALID


def test_backend_refuses_cpu_tensors():
    import pytest
    import torch

    from nifty_amd import backend as B

    with pytest.raises(RuntimeError):
        B.vdot(torch.zeros(4, dtype=torch.float64), torch.zeros(4, dtype=torch.float64))


def test_chirp_z_fallback_terminates_and_is_exact(monkeypatch):
    """Host logic of the any-length fallback in backend.py with the native transform replaced by a stand-in: the
    composition is exact, and a length whose padded convolution the planner rejects ends in NotImplementedError
    instead of re-entering the fallback (that loop once doubled its buffers until the machine ran out of memory)."""
    import numpy as np
    import torch

    from nifty_amd import backend as B

    calls = []

    def pow2_only(shape, dtype, batch=1, device=None):
        n = int(shape[-1])
        return len(shape) == 1 and n & (n - 1) == 0 and n <= 4096

    def native(x, ndim=None, inverse=False, scale=1.0):
        n = x.shape[-1]
        assert ndim == 1 and n & (n - 1) == 0, "the fallback may only call power-of-two 1-D transforms"
        calls.append(n)
        return (torch.fft.ifft(x, dim=-1, norm="forward") if inverse else torch.fft.fft(x, dim=-1)) * scale

    def plan(shape, dtype, batch=1, device=None):
        if not pow2_only(shape, dtype):
            raise NotImplementedError("stand-in planner")

    def glue(a, w, in_cols, out_cols, mode, scale=1.0, sgn=1):  # what nk_cplx_rows computes (include/niftyk.h)
        k = min(in_cols, out_cols)
        if mode == 2:
            return scale * (a.real + sgn * a.imag)[..., :k]
        src = a[..., :k].to(torch.complex128)
        out = torch.zeros(a.shape[:-1] + (out_cols,), dtype=torch.complex128)
        out[..., :k] = scale * (src if w is None else src * w[:k])
        return out

    monkeypatch.setenv("NK_BLUESTEIN", "0")  # the COMPOSITION (rows too long for the one-launch kernel's LDS take it on a GPU too)
    monkeypatch.setattr(B, "cplx_rows", glue)
    monkeypatch.setattr(B, "plan_supported", pow2_only)
    monkeypatch.setattr(B, "get_plan", plan)
    monkeypatch.setattr(B, "fftn", native)
    monkeypatch.setattr(B, "_chirps", {})
    rng = np.random.default_rng(0)
    for shape in [(11,), (10, 11), (3, 13, 6), (1,), (15,)]:
        z = torch.from_numpy(rng.normal(size=shape) + 1j * rng.normal(size=shape))
        got = B._fft_any(z, len(shape), False).numpy()
        ref = np.fft.fftn(z.numpy())
        assert np.max(np.abs(got - ref)) < 1e-12 * np.max(np.abs(ref))
        got = B._fft_any(z, len(shape), True).numpy()
        ref = np.fft.ifftn(z.numpy()) * z.numel()
        assert np.max(np.abs(got - ref)) < 1e-12 * np.max(np.abs(ref))
    n_before = len(calls)
    with pytest.raises(NotImplementedError):  # 2n-1 > 4096: the stand-in planner rejects the padded length
        B._fft_any(torch.zeros(3000, dtype=torch.complex128), 1, False)
    with pytest.raises(NotImplementedError):  # beyond the fallback's own cap, nothing is built at all
        B._fft_any(torch.zeros(20011, dtype=torch.complex128), 1, False)
    assert len(calls) == n_before and all(k[0] not in (3000, 20011) for k in B._chirps)
