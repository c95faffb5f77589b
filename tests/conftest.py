import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests are the parity tests proper and need an MI355X plus the built HIP library: on a host without them a
    plain `pytest tests` skips them instead of erroring out in every test_gpu_* file.  (On a GPU box a MISSING library is
    a failure, not a skip: the product has no fallback path, and the driver checks that the native code was loaded.)"""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except ImportError:                 # a host without torch still runs the pure-numpy oracle tests
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="needs a GPU (MI355X); run with -m gpu on the GPU box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def trained_phi4():
    """bench.py's state at the start of its timed region (one trained cycle), computed once per session on the GPU."""
    from tests import gpu_util as gu
    return gu.train_phi4_like_bench()
