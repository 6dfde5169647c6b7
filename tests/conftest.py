import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import models  # noqa: E402,F401  (registers scalable-e3-gnn_amd/ as `scalable_e3_gnn_amd`)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_meta():
    import json

    with open(os.path.join(GOLDEN, "l1tp_meta.json")) as f:
        return json.load(f)


def load_case(name):
    import numpy as np

    return np.load(os.path.join(GOLDEN, f"l1tp_{name}.npz"))
