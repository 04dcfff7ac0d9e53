import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hmm_params():
    d = np.load(os.path.join(GOLDEN, "hmm_params.npz"))   # allow_pickle=False (default)

    def get(key):
        return d[key + "_pi"], d[key + "_T"], d[key + "_E"]
    get.keys = sorted({k.rsplit("_", 1)[0] for k in d.files})
    return get


@pytest.fixture(scope="session")
def hmm_params_file():
    return np.load(os.path.join(GOLDEN, "hmm_params.npz"))


@pytest.fixture(scope="session")
def example_pairs():
    d = np.load(os.path.join(GOLDEN, "example_pairs.npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def golden_loglik():
    with open(os.path.join(GOLDEN, "loglik_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_lib
    oracle_lib.build()
    return oracle_lib


def rel_err(a, b):
    if a == b:
        return 0.0
    return abs(a - b) / max(abs(b), 1e-300)
