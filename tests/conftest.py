import os
import sys

import pytest

# The oracle's OpenMP regions (the delta map, the exact-sum adjudicator) start a team per call: with the GPU box's 256 hardware
# threads that costs milliseconds per iteration of a 128 x 128 test.  Sixteen threads are plenty for every test size.
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import cv_oracle
    cv_oracle.lib()
    return cv_oracle


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
