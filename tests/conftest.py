import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP GPU (MI355X); run with -m gpu")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def golden_params(g):
    return json.loads(str(g["params_json"]))


@pytest.fixture(scope="session")
def orc():
    import oracle

    oracle.lib()
    return oracle


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (1.0 + np.abs(b).max()))


@pytest.fixture(autouse=True)
def _seed_everything(request):
    """Scene construction uses python `random` (asset shuffle, like the reference) and torch's global generator:
    seed them per test so a failure can be reproduced."""
    import random
    import zlib

    seed = zlib.crc32(request.node.nodeid.encode()) & 0x7FFFFFFF
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    try:
        import torch

        torch.manual_seed(seed)
    except ImportError:
        pass
    yield
