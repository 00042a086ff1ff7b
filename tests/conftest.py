import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def mp_cases():
    return load_json("mpmath_cases.json")


def case_theta(c):
    """C-ABI theta vector of a golden case."""
    return np.concatenate([np.asarray(c["ls"]).ravel(), c["kv"], c["alpha"], [c["gv"], c["jitter"]]])
