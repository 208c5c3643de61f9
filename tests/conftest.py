import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _deterministic_rngs():
    """every test starts from the same host RNG state: inputs drawn without an explicit generator (torch.randn, numpy's global state)
    are then the same from run to run, so a pass on one box is a pass on the next (the chaotic post-contact comparisons in particular)"""
    import random

    import numpy as np
    random.seed(1234)
    np.random.seed(1234)
    try:
        import torch
        torch.manual_seed(1234)
    except Exception:
        pass
    yield
