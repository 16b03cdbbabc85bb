import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.hookimpl(trylast=True)       # after the -m / -k deselection: only what will actually run counts
def pytest_collection_modifyitems(config, items):
    """A `-m gpu` run on a box without a GPU must fail loudly, not pass by collecting nothing that can run: when GPU-marked tests were selected and no
    device is visible, the session stops here with an error."""
    if not any(it.get_closest_marker("gpu") for it in items):
        return
    import torch
    if not torch.cuda.is_available():
        raise pytest.UsageError(f"{sum(1 for it in items if it.get_closest_marker('gpu'))} tests marked `gpu` were selected but no GPU is visible "
                                "(run them through gpurun, or select the CPU suite with -m \"not gpu\")")


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
