import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
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
