import os
import sys

# The GPU box has 256 host cores.  The oracle's libgomp and torch's bundled OpenMP
# runtime are separate thread pools; left at their defaults they spawn 256 spinning
# threads each and starve one another.  Cap and make waits passive BEFORE either loads.
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")
os.environ.setdefault("MKL_NUM_THREADS", os.environ["OMP_NUM_THREADS"])

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    from oracle import clib
    clib.build()
