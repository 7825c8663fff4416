import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  -- before libcds_hip.so: both link the HIP runtime, and torch must be the one that loads it

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the parity tests feed the sparse library dense-uniform and random images on purpose (worst cases of the format); the
# density policy itself (cds_sparse_library_set_density_limit) is tested where it is set explicitly
os.environ.setdefault("CDS_SPARSE_DENSITY_LIMIT", "2")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # keep libcds_hip.so in step with its sources (a no-op when it is up to date; hipcc cross-compiles without a GPU)
    import shutil
    import subprocess
    if shutil.which("make") and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        r = subprocess.run(["make", "-s", "-j4", "-C", os.path.join(ROOT, "colormipsearch_amd", "csrc")], capture_output=True, text=True)
        if r.returncode != 0:
            print("WARNING: building libcds_hip.so failed:\n" + r.stdout[-2000:] + r.stderr[-2000:], file=sys.stderr)


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
