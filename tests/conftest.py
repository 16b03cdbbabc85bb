import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def pytest_collection_modifyitems(config, items):
    # a `-m gpu` run on a box without a GPU must fail loudly, not skip silently
    pass


@pytest.fixture(autouse=True)
def _default_precisions():
    """Every test starts from the library defaults (ops.MLP_PRECISION = encoder.ENCODER_PRECISION = "auto": guarded fp16 kernels for no-grad
    work, fp32 for gradients) whatever an earlier test selected."""
    from mvsnerf_amd import ops, encoder
    ops.MLP_PRECISION = "auto"
    encoder.ENCODER_PRECISION = "auto"
    yield
    ops.MLP_PRECISION = "auto"
    encoder.ENCODER_PRECISION = "auto"
