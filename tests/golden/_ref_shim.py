"""Import shim for the upstream reference (THIS container only; never on the GPU box).

The reference lives read-only under /root/reference and is not pip-installed, and it
imports ``ducc0`` unconditionally (absent here).  This shim lets ``import nifty.cl``
succeed and makes the reference fall back to its own scipy.fft / numpy.vdot path
(reference nifty/cl/ducc_dispatch.py:152-156), i.e. "the nifty.cl numpy path" that
BASELINE.json names as the parity target.

Only tests/golden/make_golden.py (fixture generator) and the optional
``-m reference`` cross-checks import this module.
"""
import importlib.metadata as _md
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("NIFTY_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "nifty", "cl"))


def load():
    """Return the reference's ``nifty.cl`` module (scipy/numpy fallback path)."""
    if not available():
        raise ImportError(f"reference tree not found under {REFERENCE_ROOT}")
    if "nifty.cl" in sys.modules:
        return sys.modules["nifty.cl"]
    orig_version = _md.version

    def version(name):
        return "9.2.0" if name == "nifty" else orig_version(name)

    _md.version = version
    if "ducc0" not in sys.modules:
        stub = types.ModuleType("ducc0")  # not a package => `import ducc0.fft` -> ImportError
        stub.misc = types.SimpleNamespace(
            resize_thread_pool=lambda n: None, available_hardware_threads=lambda: 1
        )
        sys.modules["ducc0"] = stub
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import nifty.cl as ift  # noqa: E402

    return ift
