import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session")
def dev():
    import torch
    from gaviko_amd import lib
    lib.require_device()
    return torch.device("cuda:0")


GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(autouse=True)
def _pin_torch_seed():
    """The device-side dropout epoch word starts from torch.initial_seed() (engine._seed_base: torch.manual_seed is the user's knob),
    so every test pins it: the draws of the live-dropout tests are then the same in every run and every test order."""
    import torch
    torch.manual_seed(20261004)
    yield
