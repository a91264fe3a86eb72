import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_parity():
    return np.load(os.path.join(GOLDEN, "parity.npz"))


@pytest.fixture(scope="session")
def golden_tables():
    return np.load(os.path.join(GOLDEN, "reference_tables.npz"))


@pytest.fixture(scope="session")
def golden_basis():
    return np.load(os.path.join(GOLDEN, "basis.npz"))


@pytest.fixture(scope="session")
def golden_api():
    import json
    with open(os.path.join(GOLDEN, "api_semantics.json")) as f:
        return json.load(f)
