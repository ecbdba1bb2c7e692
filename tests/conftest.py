import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests fail loudly (not skip) when selected without a GPU; they are simply
    deselected by `-m "not gpu"` on CPU-only hosts."""
    return


@pytest.fixture(autouse=True)
def _oracle_rounding_points_follow_the_product_switches():
    """the rounding-matched oracle functions (oracle/ref_cpu.py *_bf16_points) round the stored gelu' where the product does: 8-bit
    fixed point by default (functional.STORE_GELU_GRAD == 2), bf16 under AP_GELU_STORE_GRAD=1"""
    try:
        from autoprog_amd import functional as AF
        from oracle import ref_cpu as R
        R.GELU_GRAD_BITS = 8 if AF.STORE_GELU_GRAD == 2 else 16
        R.POOL_GRAD_ROUNDED = not AF.FUSE_POOL_BWD
    except Exception:
        pass
    yield
