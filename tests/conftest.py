import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(REPO, "tests", "golden", "reference_golden.npz"))


@pytest.fixture(scope="session")
def golden_nets():
    return np.load(os.path.join(REPO, "tests", "golden", "reference_nets.npz"))


@pytest.fixture(scope="session")
def mano_dict():
    from dsf_amd.assets import build_synthetic_mano
    return build_synthetic_mano(0)


@pytest.fixture(scope="session")
def oracle_hand(mano_dict):
    from oracle.hand_ref import HandModel
    return HandModel(mano_dict)
