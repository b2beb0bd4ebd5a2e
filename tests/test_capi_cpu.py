"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and
exports every symbol include/hq_solver.h declares; without a GPU it refuses to
create a context (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

import hercules_amd as ha
from hercules_amd import build as hbuild
from hercules_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    hbuild.build()
    return ha.load_library()


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    return re.findall(r"HQ_API\s+[\w\s\*]+?\b(hq\w+)\s*\(", txt)


def test_every_declared_symbol_is_exported(lib):
    names = _declared("hq_solver.h")
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(names) == sorted(capi.EXPORTS)


def test_host_library_exports(lib):
    path = hbuild.build_host()
    if path is None:
        pytest.skip("hq_host.c not present")
    h = ctypes.CDLL(path)
    names = _declared("hq_host.h")
    assert names
    for n in names:
        assert hasattr(h, n), n


def test_no_cpu_fallback(lib):
    if ha.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(ha.HqError) as ei:
        ha.Solver(np.zeros((1, 8), np.int32), np.ones((1, 4)), np.ones((8, 7)), 1e-3)
    assert "no CPU path" in str(ei.value) or "hq error -4" in str(ei.value)


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(ha.HqError):
        ha.load_library(str(tmp_path / "libhq_solver.so"))
