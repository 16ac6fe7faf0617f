import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "real-time-deepfake-speech-detection_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def sub_sd(z, prefix):
    """Arrays named '<prefix>sd.<key>' -> {key: tensor}."""
    p = prefix + "sd."
    return {k[len(p):]: torch.from_numpy(v) for k, v in z.items() if k.startswith(p)}


@pytest.fixture(scope="session")
def has_gpu():
    return torch.cuda.is_available()
