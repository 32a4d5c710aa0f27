import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """ctypes handle on the CPU restatement (test infrastructure, oracle/bwb_oracle.c)."""
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def golden():
    return GOLDEN


@pytest.fixture(scope="session")
def built():
    """Builds the product (host binaries + C-ABI HIP library) once per session."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "bwbble_amd")], check=True)
    return os.path.join(ROOT, "bwbble_amd")
