import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  -- before libcds_hip.so: both link the HIP runtime, and torch must be the one that loads it

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fixtures():
    """The reference's own test images, decoded (tests/golden/make_fixtures.py)."""
    path = os.path.join(ROOT, "tests", "golden", "cds_fixtures.npz")
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def oracle():
    from oracle import cds_oracle
    cds_oracle.lib()
    return cds_oracle
