"""solver_float = 4 in the C host side (hq_host.h): the n_t rows as the reference's -DSINGLE_PRECISION_SOLVER build sums them
(psolve.h:60-64: mass_simple, mass2_minusaM[3], mass_minusaM[3] are float fields; every `+=` / `-=` of psolve.c:3440-3471
and of the hanging nodes' mass distribution :5958-5990 rounds to float).  Checked against the oracle built with the same
define (oracle/libherc_oracle_f32.so), which the float reference's own checkpoints pin (tests/test_oracle_golden.py)."""
import numpy as np
import pytest

from hercules_amd import capi, host
from oracle import herc_oracle as ho
from tests import helpers as H

LAYERS = [(0.0, 1500.0, 300.0, 1800.0), (50.0, 3000.0, 1400.0, 2200.0), (120.0, 6000.0, 3464.0, 2700.0)]


def _exact_floats(t):
    return np.array_equal(t, t.astype(np.float32).astype(np.float64))


@pytest.mark.parametrize("shape,layers,damping", [((16, 16, 8), None, "rayleigh"), ((8, 32, 4), LAYERS, "rayleigh"),
                                                  ((16, 8, 8), LAYERS, "mass"), ((4, 4, 2), None, "none")])
def test_uniform_box_rows_are_the_float_builds(shape, layers, damping):
    nx, ny, nz = shape
    h, dt, freq = 62.5, 1e-3, 5.0
    b = host.Box(nx, ny, nz, h, dt, freq, layers=layers, damping=damping, solver_float=4)
    d = host.Box(nx, ny, nz, h, dt, freq, layers=layers, damping=damping)
    elem_ijk, lnid, node_ijk = ho.uniform_mesh(nx, ny, nz)
    edata = np.empty((len(lnid), 4), np.float32)
    edata[:, 0] = h
    edata[:, 1:] = d.material()
    et, nt = ho.solver_init(lnid, edata, ho.face_bits(elem_ijk, nx, ny, nz), len(node_ijk), dt, freq,
                            damping=ho.DAMPING_BY_NAME[damping], real=np.float32)
    assert nt.dtype == np.float32
    assert np.array_equal(b.etable, et) and np.array_equal(b.etable, d.etable)   # e_t is double in both builds (psolve.h:196-199)
    assert _exact_floats(b.ntable) and np.array_equal(b.ntable.astype(np.float32), nt)
    # not the double build's rows rounded once: sums of eight float-rounded steps
    assert not np.array_equal(b.ntable, d.ntable) and np.abs(b.ntable / d.ntable - 1).max() < 4e-7
    b.close()
    d.close()


def _multi_rank_float_oracle(elem_ticks, edata, nranks, dt, freq, far=H.C1_FAR_TICKS):
    """octor's partition restated from the global view + solver_init on every rank + the three mass exchanges, all in float
    (ho.multi_rank_init: pinned on the float reference's 8-rank stripes, tests/test_oracle_single_precision.py)."""
    m = ho.octree_mesh_from_elem_ticks(elem_ticks, far)
    parts = ho.octree_partition(m, nranks, [f // m["emin"] for f in far])
    eds = [np.ascontiguousarray(edata[p["elems"]]) for p in parts]
    fcs = [np.ascontiguousarray(m["face"][p["elems"]]) for p in parts]
    _, nts = ho.multi_rank_init(parts, eds, fcs, dt, freq, real=np.float32)
    return m, parts, nts


@pytest.mark.parametrize("nranks", [3, 8])
def test_a_box_partitions_rows_are_the_n_rank_float_builds(nranks):
    """hq_host.c, nt_rank_rows: on N ranks the float build sums every rank's elements apart and adds the sharers' rows to the
    owner's in messenger order -- other roundings than one rank's loop leaves, and a run feels them (5e-5 against 1e-6 on the
    reference's own 8-rank checkpoints, tests/test_gpu_single_precision.py).  A partition's rows, material differing from
    element to element: bit for bit the multi-rank float oracle's on every node the rank owns; the other copies hold their
    owner's row."""
    nx, ny, nz, h, dt, freq = 16, 16, 8, 62.5, 1e-3, 5.0
    kw = dict(layers=LAYERS, lateral_classes=61, lateral_amp=0.1)
    whole = host.Box(nx, ny, nz, h, dt, freq, **kw)
    edata = np.empty((len(whole.lnid), 4), np.float32)
    edata[:, 0] = h
    edata[:, 1:] = whole.material()
    m, parts, nts = _multi_rank_float_oracle(whole.node_ijk[whole.lnid].astype(np.int64) << 26, edata, nranks, dt, freq)
    assert np.array_equal(m["lnid"], whole.lnid)
    gid = {tuple(v): i for i, v in enumerate(whole.node_ijk.tolist())}
    rows = np.zeros((len(gid), 7), np.float32)
    boxes = [host.Box(nx, ny, nz, h, dt, freq, rank=r, nranks=nranks, solver_float=4, **kw) for r in range(nranks)]
    one_rank = host.Box(nx, ny, nz, h, dt, freq, solver_float=4, **kw)
    differ = 0
    for r, b in enumerate(boxes):
        own = b.owner == r
        assert _exact_floats(b.ntable) and np.array_equal(b.ntable[own].astype(np.float32), nts[r][own])
        sch = b.schedule()                                # the messenger lists in schedule_build's order (hq_host.c: hqh_share_list)
        for lst in ("c", "s"):
            exp = parts[r]["an_sched"].get(lst, [])
            assert [q for q, _ in sch[lst]] == [q for q, _ in exp]
            for (_, a), (_, e) in zip(sch[lst], exp):
                assert np.array_equal(a, e)
        g = np.array([gid[tuple(v)] for v in b.node_ijk.tolist()])
        rows[g[own]] = nts[r][own]
        differ += int((b.ntable[own] != one_rank.ntable[g[own]]).any(axis=1).sum())
    for b in boxes:                                       # every harbored copy = the owner's row
        g = np.array([gid[tuple(v)] for v in b.node_ijk.tolist()])
        assert np.array_equal(b.ntable.astype(np.float32), rows[g])
        b.close()
    assert differ > 0                                     # not the one-rank order
    one_rank.close()
    whole.close()


@pytest.mark.parametrize("name", ["c1_short", "c5_two_level", "c5_three_level", "c5_basin", "c5_gradient"])
def test_mesh_from_leaves_on_the_references_meshes(name):
    """The reference's own meshes (hanging nodes of every kind in c5_basin / c5_gradient): rows incl. the mass the hanging
    nodes hand to their anchors, bit for bit the float oracle's."""
    g = H.load(name)
    et = g["elem_ticks"]
    edge = et[:, 7, 0] - et[:, 0, 0]
    mat = g["mat_vs_vp_rho"]
    edata = np.empty((len(et), 4), np.float32)
    edata[:, 0] = (edge * (1000.0 / 2 ** 30)).astype(np.float32)
    edata[:, 1], edata[:, 2], edata[:, 3] = mat[:, 1], mat[:, 0], mat[:, 2]
    real = H.c1_problem(real=np.float32) if name == "c1_short" else H.c5_problem(name, real=np.float32)
    assert real["ntable"].dtype == np.float32
    ob = host.OctBox.from_leaves(et[:, 0, :], edge, edata, H.C1_FAR_TICKS, 1e-3, float(g["freq"]), solver_float=4)
    assert np.array_equal(ob.etable, real["etable"])
    assert _exact_floats(ob.ntable) and np.array_equal(ob.ntable.astype(np.float32), real["ntable"])
    ob.close()


@pytest.mark.parametrize("shape", [(32, 32, 4, 6), (8, 12, 2, 1)])
def test_two_level_box_rows(shape):
    nx, ny, nzf, nzc = shape
    ob = host.OctBox(nx, ny, nzf, nzc, 31.25, 1e-3, 5.0, solver_float=4)
    if shape == (32, 32, 4, 6):
        real = H.c5_problem(real=np.float32)
        assert np.array_equal(ob.lnid, real["lnid"]) and np.array_equal(ob.ntable.astype(np.float32), real["ntable"])
    assert _exact_floats(ob.ntable)
    from_leaves = host.OctBox.from_leaves(ob.node_xyz[ob.lnid[:, 0]] * (1 << 25), (ob.node_xyz[ob.lnid[:, 7], 0] -
                                          ob.node_xyz[ob.lnid[:, 0], 0]) * (1 << 25), _edata_of(ob, 31.25),
                                          (nx << 25, ny << 25, (nzf + 2 * nzc) << 25), 1e-3, 5.0, solver_float=4)
    assert np.array_equal(from_leaves.lnid, ob.lnid) and np.array_equal(from_leaves.ntable, ob.ntable)
    from_leaves.close()
    ob.close()


def _edata_of(ob, h):
    """edata of a two-level box made with the defaults of host.OctBox (top / bottom material)."""
    size = ob.node_xyz[ob.lnid[:, 7], 0] - ob.node_xyz[ob.lnid[:, 0], 0]
    ed = np.empty((ob.E, 4), np.float32)
    ed[:, 0] = (h * size).astype(np.float32)
    fine = size == 1
    ed[fine, 1:] = (3000.0, 1732.0, 2200.0)
    ed[~fine, 1:] = (6000.0, 3464.0, 2700.0)
    return ed


@pytest.mark.parametrize("nranks", [3, 8])
def test_octree_partitions_built_locally_equal_the_cut_of_the_whole_box(nranks, monkeypatch):
    """Both constructions of a partition (cut out of the whole box | from the sorted leaf keys alone) replay the N-rank float
    build's three mass exchanges (runs of ranks, the hanging nodes' shares at their owners): equal rows, bit for bit the
    multi-rank float oracle's where the rank owns the node."""
    whole = host.OctBox(32, 16, 6, 5, 31.25, 1e-3, 5.0)
    size = whole.node_xyz[whole.lnid[:, 7], 0] - whole.node_xyz[whole.lnid[:, 0], 0]
    et = whole.node_xyz[whole.lnid].astype(np.int64) << 25
    _, parts, nts = _multi_rank_float_oracle(et, _edata_of(whole, 31.25), nranks, 1e-3, 5.0, far=(32 << 25, 16 << 25, 16 << 25))
    assert (size > 1).any()
    for rank in range(nranks):
        monkeypatch.setenv("HQH_OCTBOX_LOCAL", "0")
        a = host.OctBox(32, 16, 6, 5, 31.25, 1e-3, 5.0, rank=rank, nranks=nranks, solver_float=4)
        monkeypatch.setenv("HQH_OCTBOX_LOCAL", "1")
        b = host.OctBox(32, 16, 6, 5, 31.25, 1e-3, 5.0, rank=rank, nranks=nranks, solver_float=4)
        assert np.array_equal(a.node_xyz, b.node_xyz) and np.array_equal(a.ntable, b.ntable)
        assert np.array_equal(a.lnid, parts[rank]["lnid"])
        own = a.owner == rank
        assert np.array_equal(a.ntable[own].astype(np.float32), nts[rank][own])
        a.close()
        b.close()
    whole.close()


def test_a_solver_float_that_is_neither_is_refused():
    with pytest.raises(capi.HqError):
        host.Box(4, 4, 2, 62.5, 1e-3, 5.0, solver_float=2)
    with pytest.raises(capi.HqError):
        host.OctBox(8, 8, 2, 1, 31.25, 1e-3, 5.0, solver_float=16)


def test_the_rows_as_the_float_array_a_float_reference_holds():
    """hqh_ntable_to_float: hq_desc.nTable of libhq_solver_f32.so from this library's double arrays -- exact with solver_float = 4."""
    import ctypes
    b = host.Box(8, 8, 4, 62.5, 1e-3, 5.0, layers=LAYERS, solver_float=4)
    out = np.empty(b.ntable.shape, np.float32)
    lib = host.load_library()
    assert lib.hqh_ntable_to_float(b.ntable.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(len(b.ntable)),
                                   out.ctypes.data_as(ctypes.c_void_p)) == 0
    assert np.array_equal(out.astype(np.float64), b.ntable)
    assert lib.hqh_ntable_to_float(None, ctypes.c_int64(1), out.ctypes.data_as(ctypes.c_void_p)) != 0
    b.close()


def test_partition_rows_against_the_multi_rank_float_oracle():
    """hqh_mesh_from_leaves on 8 partitions of the float reference's own two-level mesh (800 hanging nodes, shared between
    ranks): the rows a rank OWNS bit for bit the multi-rank float oracle's -- the tables the float reference's 8 ranks
    stepped to the stripes in tests/golden/c5_two_level_np8_f32.npz."""
    g = H.load("c5_two_level_np8_f32")
    base = H.load(str(g["base"]))
    et = base["elem_ticks"]
    edge = et[:, 7, 0] - et[:, 0, 0]
    mat = base["mat_vs_vp_rho"]
    edata = np.empty((len(et), 4), np.float32)
    edata[:, 0] = (edge * (1000.0 / 2 ** 30)).astype(np.float32)
    edata[:, 1], edata[:, 2], edata[:, 3] = mat[:, 1], mat[:, 0], mat[:, 2]
    pr = H.c5_np8_problem("c5_two_level_np8_f32", real=np.float32)
    one = host.OctBox.from_leaves(et[:, 0, :], edge, edata, H.C1_FAR_TICKS, 1e-3, float(base["freq"]), solver_float=4)
    differing = 0
    for r in range(8):
        b = host.OctBox.from_leaves(et[:, 0, :], edge, edata, H.C1_FAR_TICKS, 1e-3, float(base["freq"]), rank=r, nranks=8,
                                    solver_float=4)
        assert np.array_equal(b.lnid, pr["parts"][r]["lnid"])
        own = b.owner == r
        assert _exact_floats(b.ntable) and np.array_equal(b.ntable[own].astype(np.float32), pr["nts"][r][own])
        differing += int((b.ntable[own] != one.ntable[b.gid[own]]).any(axis=1).sum())
        b.close()
    one.close()
    assert differing > 0                                # one rank's order would have given other floats on the interfaces


@pytest.mark.parametrize("name,nranks", [("c5_basin", 5), ("c5_gradient", 8)])
def test_lateral_basin_partitions_against_the_multi_rank_float_oracle(name, nranks):
    """The same on the laterally refined basins (hanging nodes of every kind, anchors held through indirect sharing, material
    differing from element to element in c5_gradient)."""
    g = H.load(name)
    et = g["elem_ticks"]
    edge = et[:, 7, 0] - et[:, 0, 0]
    mat = g["mat_vs_vp_rho"]
    edata = np.empty((len(et), 4), np.float32)
    edata[:, 0] = (edge * (1000.0 / 2 ** 30)).astype(np.float32)
    edata[:, 1], edata[:, 2], edata[:, 3] = mat[:, 1], mat[:, 0], mat[:, 2]
    _, parts, nts = _multi_rank_float_oracle(et, edata, nranks, 1e-3, float(g["freq"]))
    for r in range(nranks):
        b = host.OctBox.from_leaves(et[:, 0, :], edge, edata, H.C1_FAR_TICKS, 1e-3, float(g["freq"]), rank=r, nranks=nranks,
                                    solver_float=4)
        assert np.array_equal(b.lnid, parts[r]["lnid"])
        own = b.owner == r
        assert np.array_equal(b.ntable[own].astype(np.float32), nts[r][own])
        b.close()
