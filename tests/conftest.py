import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_sessionstart(session):
    """Build the native pieces if they are missing (hipcc cross-compiles without a GPU; seconds for the oracle)."""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "quadrotor_landing_amd", "libqle_ekf.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "quadrotor_landing_amd", "csrc")], check=True)
    if not (os.path.exists(os.path.join(ROOT, "oracle", "libekf_oracle.so")) and os.path.exists(os.path.join(ROOT, "oracle", "libekf_oracle_structured.so"))):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
