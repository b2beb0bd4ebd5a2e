import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# the suite steers kernels and plans through HQ_* variables (monkeypatch.setenv): the library honours them only in a
# process that says so (hq_options.allow_env, include/hq_solver.h)
os.environ.setdefault("HQ_ALLOW_ENV", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
