import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; builds oracle/liboracle.so on first use)."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def hip():
    """The product's HIP path; fails loudly if the extension is missing or no GPU is visible."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test running without a GPU"
    import sfm_mvs_amd
    sfm_mvs_amd.lib()
    from sfm_mvs_amd import ops
    return ops
