import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib_built():
    from e2e_multi_view_matching_amd.build import build_library
    return build_library()


@pytest.fixture(scope="session")
def gpu(lib_built):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need an MI355X; there is no CPU fallback for the HIP path")
    return torch.device("cuda", 0)


@pytest.fixture
def split_always(gpu):
    """The split-operand arithmetic modes (bf16x3, f16x2) fall back to the fp32-MFMA kernels for calls below 16384
    keypoint rows; parity tests at small sizes switch that rule off so that they reach the split kernels."""
    from e2e_multi_view_matching_amd import _lib
    ctx = _lib.context(gpu)
    ctx.set_split_min_rows(0)
    yield
    ctx.set_split_min_rows(-1)
