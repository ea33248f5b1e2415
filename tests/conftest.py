import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _refresh_emulation_library():
    """Rebuild tests/emu/libnk_emu.so when a kernel header or an emulation source is newer than it (g++, ~2 min): the
    emulation runs the SAME phase functions as the GPU, so a stale library would silently test yesterday's kernels."""
    import glob
    import shutil
    import subprocess

    emu_dir = os.path.join(ROOT, "tests", "emu")
    lib = os.path.join(emu_dir, "libnk_emu.so")
    srcs = [os.path.join(emu_dir, "emu_fft.cpp"), os.path.join(emu_dir, "emu_rng.cpp")]
    deps = srcs + glob.glob(os.path.join(ROOT, "nifty_amd", "csrc", "*.h")) + [os.path.join(ROOT, "include", "niftyk.h")]
    if os.path.exists(lib) and all(os.path.getmtime(d) <= os.path.getmtime(lib) for d in deps):
        return
    if shutil.which("g++") is None:
        return
    try:  # a failed refresh must not take the whole test session down: the emulation tests then speak for themselves
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", lib + ".tmp"] + srcs)
        os.replace(lib + ".tmp", lib)
    except (subprocess.CalledProcessError, OSError) as exc:
        print(f"warning: could not rebuild {lib}: {exc!r}", file=sys.stderr)


def pytest_configure(config):
    if not os.environ.get("PYTEST_XDIST_WORKER"):
        _refresh_emulation_library()
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: cross-check against /root/reference (build container only)")
    # pytest-timeout (in this image) enforces it; without the plugin the marker is inert -- the multi-process tests bound
    # their children with subprocess timeouts of their own
    config.addinivalue_line("markers", "timeout(seconds): upper bound of a test's run time (pytest-timeout, optional)")


def pytest_collection_modifyitems(config, items):
    import torch

    has_gpu = torch.cuda.is_available()
    skip_gpu = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords and not has_gpu:
            item.add_marker(skip_gpu)
