import os
import sys
from pathlib import Path

import pytest

# The oracle's C restatement uses OpenMP (one parallel region per column of its dense Cholesky): on the GPU box's 256-thread host an
# unbounded team turns a one-second factorisation into minutes.  Bound it before libgomp starts (the library is loaded lazily).
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
