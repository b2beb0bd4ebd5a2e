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
# several tests build eight partitions side by side in threads, each of whose C calls opens OpenMP regions: idle workers
# must sleep, not spin, or a box with few cores spends its time in barriers (libgomp: hq_host; libomp: hq_solver)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("GOMP_SPINCOUNT", "0")
os.environ.setdefault("KMP_BLOCKTIME", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# tests/test_gpu_parity.py runs every case twice (fixture brick_mode: as shipped / HQ_NO_BRICKS=1).  For the SCATTER
# variant the switch selects nothing -- it has neither bricks nor patches -- and a handful of cases never reach a kernel
# that knows it: their second run is dropped at collection (round 6: the suite must fit the driver's time limit).
_MODE_BLIND = ("test_rccl_binding_on_a_communicator_of_one", "test_create_refuses_a_table_that_is_not_rayleigh_proportional",
               "test_phase_force_and_update_match_reference_loops")


def pytest_collection_modifyitems(config, items):
    keep, drop = [], []
    for it in items:
        cs = getattr(it, "callspec", None)
        if it.fspath.basename == "test_gpu_parity.py" and cs is not None and cs.params.get("brick_mode") == "patches-only":
            if cs.params.get("variant") == 1 or it.originalname in _MODE_BLIND:      # 1 = HQ_VARIANT_SCATTER
                drop.append(it)
                continue
        keep.append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep
