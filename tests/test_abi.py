"""The C-ABI library loads and exports every symbol include/niftyk.h declares (no compute calls: CPU box)."""
import ctypes
import os
import re

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
