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


def test_single_precision_build_exports_the_same_interface(lib):
    """libhq_solver_f32.so: the same sources with -DHQ_SINGLE_PRECISION_SOLVER (hq_real = float, the reference's
    -DSINGLE_PRECISION_SOLVER psolve.h:60-64): every declared symbol, sizeof(hq_real) = 4; the default library says 8."""
    f32 = capi.load_library(precision="f32")
    for n in _declared("hq_solver.h"):
        assert hasattr(f32, n), n
    assert f32.hq_real_bytes() == 4 and lib.hq_real_bytes() == 8
    assert f32.hq_abi_version() == lib.hq_abi_version()
    if ha.device_count() == 0:          # no CPU path in this build either
        with pytest.raises(capi.HqError):
            ha.Solver(np.zeros((1, 8), np.int32), np.ones((1, 4)), np.ones((8, 7)), 1e-3, precision="f32")


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


def test_options_struct_matches_the_header_and_initialises_to_defaults(lib):
    """hq_options (ABI 6): the ctypes mirror has the header's fields in the header's order, hq_options_init writes -1
    ("library default") into every one of them and never past the size it is given."""
    txt = open(os.path.join(ROOT, "include", "hq_solver.h")).read()
    body = txt[txt.index("typedef struct {\n    uint64_t size;"):txt.index("} hq_options;")]
    fields = re.findall(r"^\s+(?:int32_t|double|uint64_t)\s+(\w+);", body, re.M)
    assert fields == [n for n, _ in capi.Options._fields_]
    assert re.search(r"#define HQ_ABI_VERSION 6", txt) and lib.hq_abi_version() == 6
    o = capi.Options()
    assert o.size == ctypes.sizeof(o) and all(v == -1 for v in o.as_dict().values())
    o = capi.Options(brick_cz=16, ipc_timeout_ms=250.0)
    assert o.brick_cz == 16 and o.ipc_timeout_ms == 250.0 and o.no_bricks == -1
    with pytest.raises(TypeError):
        capi.Options(no_such_field=1)
    # a struct too short to hold even its size field is left alone (round-5 advisor)
    tiny = (ctypes.c_char * 8)(*([b"\x55"] * 8))
    lib.hq_options_init(tiny, ctypes.c_uint64(4))
    assert bytes(tiny) == b"\x55" * 8
    # an older client with a shorter struct: nothing is written past its size
    raw = (ctypes.c_char * ctypes.sizeof(capi.Options))(*([b"\x55"] * ctypes.sizeof(capi.Options)))
    lib.hq_options_init(raw, ctypes.c_uint64(24))
    assert bytes(raw)[24:] == b"\x55" * (ctypes.sizeof(capi.Options) - 24)
    assert int.from_bytes(bytes(raw)[:8], "little") == 24 and bytes(raw)[8:24] == b"\xff" * 16


def test_create_opts_without_a_device_fails_like_create(lib):
    if ha.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(ha.HqError):
        ha.Solver(np.zeros((1, 8), np.int32), np.ones((1, 4)), np.ones((8, 7)), 1e-3, options={"no_bricks": 1})


def test_info_struct_matches_the_header():
    """hq_info (ABI 6: + brick_stream and the device-side phase split t_*_us): the ctypes mirror has the header's fields."""
    txt = open(os.path.join(ROOT, "include", "hq_solver.h")).read()
    body = txt[txt.index("typedef struct {\n    int32_t variant;"):txt.index("} hq_info;")]
    fields = re.findall(r"^\s+(?:int32_t|int64_t|double)\s+(\w+);", body, re.M)
    assert fields == [n for n, _ in capi._Info._fields_]
    assert fields[-5:] == ["t_step_us", "t_shell_us", "t_interior_us", "t_chain_us", "t_chain_exposed_us"]


def test_rccl_is_bound_through_its_own_header():
    """The RCCL entry points, the id's size and the datatype enumerators come from <rccl/rccl.h> at COMPILE time
    (decltype of the header's own prototypes; only the library is looked up at run time): no hand-declared prototype, no
    hard-coded enum value is left in the engine (round-5 review 2b) -- and this image's header still says what the
    C-ABI promises (128-byte id)."""
    src = open(os.path.join(ROOT, "hercules_amd", "csrc", "hq_engine.hip")).read()
    assert "#include <rccl/rccl.h>" in src
    for fn in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclSend", "ncclRecv", "ncclGroupStart",
               "ncclGroupEnd", "ncclGetErrorString"):
        assert "decltype(&%s)" % fn in src, fn
    assert not re.search(r"HQ_NCCL_(DOUBLE|INT64)\s*=\s*\d", src)
    assert "static_assert(sizeof(hq_nccl_id) == 128" in src
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if os.path.exists(hdr):
        h = open(hdr).read()
        assert re.search(r"#define NCCL_UNIQUE_ID_BYTES 128", h) and re.search(r"ncclFloat64\s*=\s*8", h)
