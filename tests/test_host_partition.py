"""Host logic of the product's C side (hercules_amd/csrc/hq_host.c), no GPU:
mesh order, solver_init constants, block partition, node ownership, schedules,
point source and stations -- against the oracle and the reference's own
8-rank run (tests/golden/c1_np8.npz)."""
import os

import numpy as np
import pytest

from hercules_amd import host
from oracle import herc_oracle as ho
from tests import helpers as H


def _oracle_box(nx, ny, nz, h, dt, freq, layers=None, damping=ho.DAMP_RAYLEIGH):
    elem_ijk, lnid, node_ijk = ho.uniform_mesh(nx, ny, nz)
    edata = np.empty((len(lnid), 4), np.float32)
    edata[:, 0] = h
    if layers is None:
        edata[:, 1:] = (6000.0, 3464.0, 2700.0)
    else:
        zc = (elem_ijk[:, 2] + 0.5) * h
        for ztop, vp, vs, rho in layers:
            m = zc >= ztop
            edata[m, 1], edata[m, 2], edata[m, 3] = vp, vs, rho
    et, nt = ho.solver_init(lnid, edata, ho.face_bits(elem_ijk, nx, ny, nz), len(node_ijk), dt, freq, damping=damping)
    return elem_ijk, lnid, node_ijk, et, nt


@pytest.mark.parametrize("shape,layers,damping", [
    ((16, 16, 8), None, "rayleigh"),
    ((8, 32, 4), [(0.0, 1500.0, 300.0, 1800.0), (50.0, 3000.0, 1400.0, 2200.0), (120.0, 6000.0, 3464.0, 2700.0)], "rayleigh"),
    ((16, 8, 8), None, "mass"),
    ((4, 4, 2), None, "none"),
])
def test_single_partition_equals_oracle_bitwise(shape, layers, damping):
    nx, ny, nz = shape
    h, dt, freq = 62.5, 1e-3, 5.0
    b = host.Box(nx, ny, nz, h, dt, freq, layers=layers, damping=damping)
    elem_ijk, lnid, node_ijk, et, nt = _oracle_box(nx, ny, nz, h, dt, freq, layers, ho.DAMPING_BY_NAME[damping])
    assert np.array_equal(b.lnid, lnid)
    assert np.array_equal(b.node_ijk, node_ijk)
    assert np.array_equal(b.etable, et)          # float-evaluated mu/lambda/zeta reproduced (psolve.c:3242-3409)
    assert np.array_equal(b.ntable, nt)          # gather form == the reference's scatter loop, same order
    assert b.info["nowned"] == b.info["nharbored"] == len(node_ijk)
    b.close()


def test_partitions_match_reference_8_rank_run():
    """Elements per rank and the per-rank harbored-node ORDER are octor's: the
    reference's own 8-rank checkpoints line up with our local numbering."""
    g = H.load("c1_np8")
    p = H.c1_problem()
    gidx = {tuple(v): i for i, v in enumerate(p["node_ijk"].tolist())}
    edge = 1 << 26
    # single-rank oracle state at the checkpoint steps
    g1 = H.load("c1_short")
    tm1, tm2 = np.zeros((p["N"], 3)), np.zeros((p["N"], 3))
    done = 0
    states = {}
    for step in g["ckpt_steps"]:
        ho.solver_run(p["lnid"], p["etable"], p["ntable"], tm1, tm2, done, int(step) - done, p["dt"],
                      loaded_lnid=g1["loaded_lnid"], forces=g1["forces"])
        done = int(step)
        states[done] = (tm1.copy(), tm2.copy())
    for r in range(8):
        b = host.Box(16, 16, 8, 62.5, 1e-3, 5.0, rank=r, nranks=8)
        ref_elems = g["elem_ticks_%d" % r]
        assert b.info["lenum"] == len(ref_elems) == 256
        mine = (b.node_ijk[b.lnid].astype(np.int64) * edge)
        assert np.array_equal(mine, ref_elems.astype(np.int64))
        m = np.array([gidx[tuple(v)] for v in b.node_ijk.tolist()])
        for step in g["ckpt_steps"]:
            ref_tm2 = g["ckpt%d_tm2_%d" % (int(step), r)][:len(m)]
            ref_tm1 = g["ckpt%d_tm1_%d" % (int(step), r)][:len(m)]
            o1, o2 = states[int(step)]
            scale = np.abs(o2).max()
            # 1-rank vs 8-rank reference runs differ by rounding only
            assert np.abs(ref_tm1 - o2[m]).max() <= 1e-12 * scale
            assert np.abs(ref_tm2 - o1[m]).max() <= 1e-12 * scale
        b.close()


@pytest.mark.parametrize("shape,nranks", [((16, 16, 8), 8), ((32, 16, 16), 2), ((8, 8, 8), 5), ((16, 8, 4), 3)])
def test_ownership_and_schedules_are_consistent(shape, nranks):
    nx, ny, nz = shape
    boxes = [host.Box(nx, ny, nz, 10.0, 1e-4, 50.0, rank=r, nranks=nranks) for r in range(nranks)]
    elem_ijk, lnid, node_ijk = ho.uniform_mesh(nx, ny, nz)
    E = len(lnid)
    gkey = lambda ijk: (ijk[:, 2].astype(np.int64) * (ny + 1) + ijk[:, 1]) * (nx + 1) + ijk[:, 0]
    owner_global = np.full(len(node_ijk), -1)
    gid_of = {int(k): i for i, k in enumerate(gkey(node_ijk))}
    off = 0
    for r, b in enumerate(boxes):
        lo, hi = r * E // nranks, (r + 1) * E // nranks            # BLOCK_LOW / BLOCK_HIGH (octor.c:4939-4944)
        assert b.info["lenum"] == hi - lo
        assert np.array_equal(b.node_ijk[b.lnid], node_ijk[lnid[lo:hi]])
        ids = np.array([gid_of[int(k)] for k in gkey(b.node_ijk)])
        assert np.all(np.diff(ids) > 0)                             # local order == global Z-order
        for n, o in zip(ids, b.owner):
            assert owner_global[n] in (-1, o)
            owner_global[n] = o
        off += hi - lo
    assert np.all(owner_global >= 0)
    # owner = rank of the element whose lower-left corner is the (clamped) node (octor.c:5466-5475)
    eidx = {tuple(v): i for i, v in enumerate(elem_ijk.tolist())}
    for n, ijk in enumerate(node_ijk):
        e = eidx[(min(ijk[0], nx - 1), min(ijk[1], ny - 1), min(ijk[2], nz - 1))]
        assert owner_global[n] == ((e + 1) * nranks - 1) // E
    # schedules: my c-list to q mirrors q's s-list to me, node for node
    sched = [b.schedule() for b in boxes]
    for r in range(nranks):
        for q, mapping in sched[r]["c"]:
            other = dict(sched[q]["s"])
            assert r in other and len(other[r]) == len(mapping)
            a = gkey(boxes[r].node_ijk[mapping])
            c = gkey(boxes[q].node_ijk[other[r]])
            assert np.array_equal(a, c)
            assert np.all(boxes[r].owner[mapping] == q)
        for q, mapping in sched[r]["s"]:
            assert np.all(boxes[r].owner[mapping] == r)
            assert r in dict(sched[q]["c"])
    # every non-owned harbored node is in exactly one c-list
    for r, b in enumerate(boxes):
        listed = np.concatenate([m for _, m in sched[r]["c"]]) if sched[r]["c"] else np.zeros(0, int)
        assert sorted(listed.tolist()) == np.nonzero(b.owner != r)[0].tolist()
    for b in boxes:
        b.close()


def test_point_source_matches_reference_force_file():
    """source_initnodalforce: same loaded nodes and the same nodal pattern as the
    reference's force_process.0 (up to the scalar time function)."""
    g = H.load("c1_short")
    b = host.Box(16, 16, 8, 62.5, 1e-3, 5.0)
    ids, pat = b.point_source(500.0, 500.0, 100.0, 0.0, 90.0, 0.0)
    assert np.array_equal(ids, g["loaded_lnid"])
    F = g["forces"][100]
    s = F[np.unravel_index(np.abs(F).argmax(), F.shape)] / pat[np.unravel_index(np.abs(F).argmax(), F.shape)]
    assert np.abs(F - s * pat).max() <= 1e-12 * np.abs(F).max()
    b.close()


def test_stations_match_oracle_weights():
    b = host.Box(16, 16, 8, 62.5, 1e-3, 5.0)
    pts = H.C1_STATIONS + [(33.0, 977.0, 412.5), (1000.0, 1000.0, 500.0)]
    ids, phi, mine = b.stations(pts)
    oi, op = ho.station_weights(pts, 62.5, 16, 16, 8, b.lnid, ho.uniform_mesh(16, 16, 8)[0])
    assert np.array_equal(ids, oi) and np.allclose(phi, op, rtol=0, atol=1e-15) and mine.all()
    assert np.allclose(phi.sum(axis=1), 1.0)
    b.close()


def test_force_file_format_roundtrip(tmp_path):
    """force_process.<rank> layout (quakesource.c:2453-2466): a file assembled from the
    reference's payload reads back exactly; windows past the end read as zero."""
    g = H.load("c1_short")
    ids, F = g["loaded_lnid"], g["forces"]
    raw = np.int32(len(ids)).tobytes() + ids.astype("<i4").tobytes() + F.astype("<f8").tobytes()
    ref = tmp_path / "force_process.0"
    ref.write_bytes(raw)
    got_ids, nsteps = host.forcefile_info(str(ref))
    assert np.array_equal(got_ids, ids) and nsteps == F.shape[0]
    assert np.array_equal(host.forcefile_read(str(ref), len(ids), 0, nsteps), F)
    w = host.forcefile_read(str(ref), len(ids), nsteps - 3, 8)
    assert np.array_equal(w[:3], F[-3:]) and not w[3:].any()
    mine = tmp_path / "mine"
    host.forcefile_write(str(mine), ids, F)
    assert mine.read_bytes() == raw


def test_station_line_format_matches_reference_text():
    """psolve.c:6729-6731 prints "\\n%10.6f % 8e % 8e % 8e"."""
    line = host.station_format(0.002, [-1.234661e-02, -1.234661e-02, 7.560121e-19])
    assert line == "\n  0.002000 -1.234661e-02 -1.234661e-02  7.560121e-19"


@pytest.mark.parametrize("shape", [(32, 32, 4, 6), (8, 12, 2, 1), (16, 8, 6, 3)])
def test_two_level_box_of_the_c_host_is_octors(shape):
    """hqh_octbox_create (C host) against tests/helpers.two_level_mesh, the construction
    pinned bit-for-bit to the mesh the REAL reference generated (c5_two_level): element
    order, node order, hanging-node table with anchor order, eTable, nTable incl. the mass
    distribution -- all bitwise."""
    nx, ny, nzf, nzc = shape
    ob = host.OctBox(nx, ny, nzf, nzc, 31.25, 1e-3, 5.0)
    ref = H.two_level_mesh(nx, ny, nzf, nzc)
    assert ob.E == ref["E"] and ob.N == ref["N"]
    assert np.array_equal(ob.lnid, ref["lnid"])
    assert np.array_equal(ob.node_xyz, ref["node_q"])
    for a, b in zip(ob.dangling, ref["dangling"]):
        assert np.array_equal(a, b)
    assert np.array_equal(ob.etable, ref["etable"])
    assert np.array_equal(ob.ntable, ref["ntable"])
    if shape == (32, 32, 4, 6):
        real = H.c5_problem()
        assert np.array_equal(ob.lnid, real["lnid"]) and np.array_equal(ob.ntable, real["ntable"])
    ob.close()


def test_plane_geometry_of_the_c_host_equals_the_restatement():
    """hqh_plane_points / hqh_domain_coords against the oracle's restatement (itself pinned on the
    reference's plane files, test_oracle_golden.py), incl. the specs of the golden planes."""
    from hercules_amd import host
    g = H.load("c1_planes")
    lonc, latc = g["surface_corners_lon_lat"][:, 0], g["surface_corners_lon_lat"][:, 1]
    for spec in list(g["plane_specs"]) + [np.array([10.0, 20.0, 5.0, 3.0, 4, 2.5, 3, 123.0, 37.0])]:
        lat, lon, depth, ds, ns, dd, nd, strike, dip = spec
        x, y = host.domain_coords(lon, lat, lonc, latc, 1000.0, 1000.0)
        xo, yo = ho.domain_coords_linearinterp(lon, lat, lonc, latc, 1000.0, 1000.0)
        assert (x, y) == (xo, yo)
        a = host.plane_points((x, y, depth), ds, int(ns), dd, int(nd), strike, dip)
        b = ho.plane_points((xo, yo, depth), ds, int(ns), dd, int(nd), strike, dip)
        assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max())
    # a skewed quadrilateral: the corner itself maps to the domain corner
    lonq, latq = [23.1997, 23.2115, 23.2118, 23.2000], [40.7756, 40.7753, 40.7843, 40.7846]
    x, y = host.domain_coords(lonq[2], latq[2], lonq, latq, 1000.0, 2000.0)
    xo, yo = ho.domain_coords_linearinterp(lonq[2], latq[2], lonq, latq, 2000.0, 1000.0)
    assert abs(x - xo) < 1e-9 and abs(y - yo) < 1e-9


@pytest.mark.parametrize("shape,nranks", [((16, 8, 6, 3), 2), ((16, 8, 6, 3), 5), ((8, 12, 2, 1), 3),
                                          ((32, 32, 4, 6), 8)])
def test_two_level_box_partitions_of_the_c_host_are_octors(shape, nranks):
    """hqh_octbox_create with nranks > 1 against ho.octree_partition -- the restatement of octor's
    per-rank tables that reproduces the REAL reference's 8-rank run of this very mesh
    (c5_two_level_np8): harbored node sets, local connectivity, owners, the dnodeTable of owned
    hanging nodes, and both schedules, all exactly; nTable = the whole box's (summed) rows."""
    nx, ny, nzf, nzc = shape
    ref = H.two_level_mesh(nx, ny, nzf, nzc)
    parts = ho.octree_partition(ref["mesh"], nranks, (nx, ny, nzf + 2 * nzc))
    whole = host.OctBox(nx, ny, nzf, nzc, 31.25, 1e-3, 5.0)
    g_nt, g_et = whole.ntable.copy(), whole.etable.copy()
    whole.close()
    for r in range(nranks):
        ob = host.OctBox(nx, ny, nzf, nzc, 31.25, 1e-3, 5.0, rank=r, nranks=nranks)
        p = parts[r]
        assert np.array_equal(ob.gid, p["nodes"])
        assert np.array_equal(ob.lnid, p["lnid"])
        assert np.array_equal(ob.owner, p["owner"])
        assert np.array_equal(ob.node_xyz, ref["node_q"][p["nodes"]])
        for a, b in zip(ob.dangling, p["dangling"]):
            assert np.array_equal(a, b)
        assert np.array_equal(ob.etable, g_et[p["elems"]])
        assert np.array_equal(ob.ntable, g_nt[p["nodes"]])
        sch = ob.schedules()
        for kind, key in (("an", "an_sched"), ("dn", "dn_sched")):
            for lst in ("c", "s"):
                got, exp = sch[kind][lst], p[key].get(lst, [])
                assert [q for q, _ in got] == [q for q, _ in exp]
                for (_, a), (_, b) in zip(got, exp):
                    assert np.array_equal(a, b)
        ob.close()


def test_three_level_box_of_the_c_host_is_the_references_mesh():
    """hqh_octbox_create_levels on the model of tests/golden/c5_three_level (2 layers of 62.5 m
    elements, 1 of 125 m, 1 of 250 m; three materials that take every branch of mu_and_lambda):
    connectivity, node order, hanging-node table incl. anchor order, eTable, nTable -- all equal to
    the mesh the REAL reference generated; and its partitions to octor's per-rank tables."""
    real = H.c5_problem("c5_three_level")
    g = real["golden"]
    mats = {float(r[0]): (float(r[1]), float(r[0]), float(r[2])) for r in np.unique(g["mat_vs_vp_rho"], axis=0)}
    levels = [(2,) + mats[150.0], (1,) + mats[2000.0], (1,) + mats[3464.0]]
    ob = host.OctBox(16, 16, 0, 0, 62.5, 1e-3, float(g["freq"]), levels=levels)
    assert ob.E == real["E"] == int(g["total_elements"]) and ob.N == real["N"] == int(g["total_nodes"])
    assert ob.ldnnum == int(g["total_dangling"])
    assert np.array_equal(ob.lnid, real["lnid"])
    assert np.array_equal(ob.node_xyz, real["node_q"])
    for a, b in zip(ob.dangling, real["dangling"]):
        assert np.array_equal(a, b)
    assert np.array_equal(ob.etable, real["etable"])
    assert np.array_equal(ob.ntable, real["ntable"])
    g_nt = ob.ntable.copy()
    ob.close()
    m = ho.octree_mesh_from_elem_ticks(g["elem_ticks"], H.C1_FAR_TICKS)
    for nranks in (3, 8):
        parts = ho.octree_partition(m, nranks, (16, 16, 8))
        for r in range(nranks):
            ob = host.OctBox(16, 16, 0, 0, 62.5, 1e-3, float(g["freq"]), levels=levels, rank=r, nranks=nranks)
            p = parts[r]
            assert np.array_equal(ob.gid, p["nodes"]) and np.array_equal(ob.lnid, p["lnid"])
            assert np.array_equal(ob.owner, p["owner"])
            for a, b in zip(ob.dangling, p["dangling"]):
                assert np.array_equal(a, b)
            assert np.array_equal(ob.ntable, g_nt[p["nodes"]])
            sch = ob.schedules()
            for kind, key in (("an", "an_sched"), ("dn", "dn_sched")):
                for lst in ("c", "s"):
                    got, exp = sch[kind][lst], p[key].get(lst, [])
                    assert [q for q, _ in got] == [q for q, _ in exp]
                    for (_, a), (_, b) in zip(got, exp):
                        assert np.array_equal(a, b)
            ob.close()


def test_layered_model_column_and_mesh_are_the_references():
    """hqh_layered_column (Vs rule on the minimum-Vs sample + 2:1 balance) and the box built from it
    against the mesh the REAL reference made of the same layered model (tests/golden/c5_layered:
    three materials, edges of 31.25 / 62.5 / 125 m after balancing, two materials inside one level)."""
    g = H.load("c5_layered")
    model = [(0.0, 800.0, 200.0, 1700.0), (62.5, 1500.0, 450.0, 2000.0), (187.5, 2600.0, 1200.0, 2300.0)]
    col = host.layered_column(model, 500.0, 1, 8 * float(g["freq"]), vscut=100.0)
    et = g["elem_ticks"]
    tick = 1000.0 / 2 ** 30
    first = (et[:, 0, 0] == 0) & (et[:, 0, 1] == 0)
    ref = sorted((float(et[e, 0, 2]) * tick, float(et[e, 7, 0] - et[e, 0, 0]) * tick) + tuple(g["mat_vs_vp_rho"][e])
                 for e in np.nonzero(first)[0])
    assert len(col) == len(ref)
    z = 0.0
    for (edge, vp, vs, rho), (z0, e0, rvs, rvp, rrho) in zip(col, ref):
        assert (z, edge, vp, vs, rho) == (z0, e0, rvp, rvs, rrho)
        z += edge
    h, levels = host.levels_from_column(col)
    assert h == 31.25 and [n for n, _ in levels] == [2, 3, 2]
    real = H.c5_problem("c5_layered")
    ob = host.OctBox(32, 32, 0, 0, h, 1e-3, float(g["freq"]), levels=levels)
    assert ob.E == int(g["total_elements"]) and ob.N == int(g["total_nodes"]) and ob.ldnnum == int(g["total_dangling"])
    assert np.array_equal(ob.lnid, real["lnid"]) and np.array_equal(ob.node_xyz, real["node_q"])
    for a, b in zip(ob.dangling, real["dangling"]):
        assert np.array_equal(a, b)
    assert np.array_equal(ob.etable, real["etable"]) and np.array_equal(ob.ntable, real["ntable"])
    ob.close()


def test_mesh_etree_reader_and_mesh_from_leaves(tmp_path):
    """The mesh.e database the REAL reference wrote for the layered model (tests/golden/c5_layered,
    mesh_output psolve.c:2361-2562) read by the C host (hqh_etree_read: etree header, B-tree pages,
    locational keys): every element's corner, level, global node ids and edata; then the mesh tables
    built from those leaves alone (hqh_mesh_from_leaves) -- connectivity, node order, dnodeTable
    with anchor order, eTable, nTable -- equal to the restatement pinned on the reference's run."""
    import bz2
    g = H.load("c5_layered")
    real = H.c5_problem("c5_layered")
    path = tmp_path / "mesh.e"
    path.write_bytes(bz2.decompress(g["mesh_e_bz2"].tobytes()))
    ticks, level, vals = host.etree_read(str(path))
    nid, edata = host.mesh_payload(vals)
    et = g["elem_ticks"]
    assert len(ticks) == real["E"] == 2944
    assert np.array_equal(ticks.astype(np.int64), et[:, 0, :])
    edge = (et[:, 7, 0] - et[:, 0, 0])
    assert np.array_equal(np.uint64(1) << (30 - level.astype(np.uint64)), edge.astype(np.uint64))   # octor PIXELLEVEL = 30
    assert np.array_equal(nid, real["lnid"])                       # global node id = Z-order rank
    assert np.array_equal(edata[:, 1:], g["mat_vs_vp_rho"][:, [1, 0, 2]])
    assert np.array_equal(edata[:, 0], (edge * (1000.0 / 2 ** 30)).astype(np.float32))
    ob = host.OctBox.from_leaves(ticks, edge, edata, H.C1_FAR_TICKS, 1e-3, float(g["freq"]))
    assert ob.E == real["E"] and ob.N == real["N"] and ob.ldnnum == len(real["dangling"][0])
    assert np.array_equal(ob.lnid, real["lnid"]) and np.array_equal(ob.node_xyz, real["node_q"])
    for a, b in zip(ob.dangling, real["dangling"]):
        assert np.array_equal(a, b)
    assert np.array_equal(ob.etable, real["etable"]) and np.array_equal(ob.ntable, real["ntable"])
    ob.close()


@pytest.mark.parametrize("name", ["c1_short", "c5_two_level", "c5_three_level", "c5_basin", "c5_gradient"])
def test_mesh_from_leaves_on_the_references_other_meshes(name):
    """hqh_mesh_from_leaves on the element dumps of the reference's uniform, two-level and
    three-level meshes."""
    g = H.load(name)
    et = g["elem_ticks"]
    edge = et[:, 7, 0] - et[:, 0, 0]
    mat = g["mat_vs_vp_rho"]
    edata = np.empty((len(et), 4), np.float32)
    edata[:, 0] = (edge * (1000.0 / 2 ** 30)).astype(np.float32)
    edata[:, 1], edata[:, 2], edata[:, 3] = mat[:, 1], mat[:, 0], mat[:, 2]
    real = H.c1_problem() if name == "c1_short" else H.c5_problem(name)
    ob = host.OctBox.from_leaves(et[:, 0, :], edge, edata, H.C1_FAR_TICKS, 1e-3, float(g["freq"]))
    assert np.array_equal(ob.lnid, real["lnid"])
    assert np.array_equal(ob.etable, real["etable"]) and np.array_equal(ob.ntable, real["ntable"])
    if name != "c1_short":
        for a, b in zip(ob.dangling, real["dangling"]):
            assert np.array_equal(a, b)
    else:
        assert ob.ldnnum == 0
    ob.close()


@pytest.mark.parametrize("name,nranks", [("c5_layered", 8), ("c5_three_level", 5), ("c5_two_level", 8),
                                         ("c5_basin", 8), ("c5_basin", 5), ("c5_basin", 3)])
def test_mesh_from_leaves_partitions_are_octors(name, nranks):
    """hqh_mesh_from_leaves with nranks > 1 (general octrees: the leaf containing a node is found
    by Z-order search) against ho.octree_partition, for the reference's own meshes; the two-level
    one on 8 ranks is the case the REAL reference ran (c5_two_level_np8)."""
    g = H.load(name)
    et = g["elem_ticks"]
    edge = et[:, 7, 0] - et[:, 0, 0]
    mat = g["mat_vs_vp_rho"]
    edata = np.empty((len(et), 4), np.float32)
    edata[:, 0] = (edge * (1000.0 / 2 ** 30)).astype(np.float32)
    edata[:, 1], edata[:, 2], edata[:, 3] = mat[:, 1], mat[:, 0], mat[:, 2]
    m = ho.octree_mesh_from_elem_ticks(et, H.C1_FAR_TICKS)
    far_q = [f // m["emin"] for f in H.C1_FAR_TICKS]
    parts = ho.octree_partition(m, nranks, far_q)
    whole = host.OctBox.from_leaves(et[:, 0, :], edge, edata, H.C1_FAR_TICKS, 1e-3, float(g["freq"]))
    g_nt = whole.ntable.copy()
    whole.close()
    for r in range(nranks):
        ob = host.OctBox.from_leaves(et[:, 0, :], edge, edata, H.C1_FAR_TICKS, 1e-3, float(g["freq"]),
                                     rank=r, nranks=nranks)
        p = parts[r]
        assert np.array_equal(ob.gid, p["nodes"]) and np.array_equal(ob.lnid, p["lnid"])
        assert np.array_equal(ob.owner, p["owner"])
        for a, b in zip(ob.dangling, p["dangling"]):
            assert np.array_equal(a, b)
        assert np.array_equal(ob.ntable, g_nt[p["nodes"]])
        sch = ob.schedules()
        for kind, key in (("an", "an_sched"), ("dn", "dn_sched")):
            for lst in ("c", "s"):
                got, exp = sch[kind][lst], p[key].get(lst, [])
                assert [q for q, _ in got] == [q for q, _ in exp]
                for (_, a), (_, b) in zip(got, exp):
                    assert np.array_equal(a, b)
        ob.close()


@pytest.mark.parametrize("name", ["c5_basin", "c5_gradient", "c5_two_level", "c5_three_level", "c5_layered"])
def test_octree_generate_makes_the_references_meshes(name):
    """hqh_octree_generate -- octor_newtree / refinetree (Vs rule on setrec's 27-sample record) / balancetree (2:1 across
    faces and edges) restated on per-level bitmaps -- from the material model alone: leaf for leaf (corner, edge,
    pre-order) and record for record (Vp, Vs, rho incl. the vscut adjustment) the mesh the REAL reference made;
    c5_basin is the laterally refined one (x-, y-, z-normal interfaces, staircase corners after balancing)."""
    g = H.load(name)
    vp, vs, rho, cell = H.cvm_grid(name)
    spec = H.CVM_MODELS[name]
    ticks, edge, edata, far, ticksize = host.octree_generate(vp, vs, rho, cell, (1000.0, 1000.0, 500.0),
                                                             spec["freq"] * 8, spec["vscut"])    # simulation_node_per_wavelength = 8 (numerical.in:24)
    et = g["elem_ticks"]
    assert far == H.C1_FAR_TICKS and ticksize == 1000.0 / 2 ** 30
    assert len(ticks) == len(et) == int(g["total_elements"])
    assert np.array_equal(ticks.astype(np.int64), et[:, 0, :]) and np.array_equal(edge.astype(np.int64), et[:, 7, 0] - et[:, 0, 0])
    assert np.array_equal(edata[:, 1:], g["mat_vs_vp_rho"][:, [1, 0, 2]])
    assert np.array_equal(edata[:, 0], (edge * (1000.0 / 2 ** 30)).astype(np.float32))


MAKE_CVM = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "make_cvm")


def _cvm_args(name):
    """oracle/make_cvm's arguments for a model of tests/helpers.CVM_MODELS (as tests/golden/make_golden.py passed them)."""
    spec = H.CVM_MODELS[name]
    if "layers" in spec:
        a = ["layers", len(spec["layers"])]
        for row in spec["layers"]:
            a += list(row)
        return [str(v) for v in a]
    a = ["regions", 4, *spec["background"], len(spec["regions"])]
    for r in spec["regions"]:
        a += list(r)
    return [str(v) for v in a]


@pytest.mark.skipif(not os.path.exists(MAKE_CVM), reason="oracle/_ref/make_cvm (oracle/build_ref.sh, the reference's etree + cvm libraries) is not built")
@pytest.mark.parametrize("name", ["c5_basin", "c5_gradient", "c5_three_level"])
def test_from_the_cvm_database_to_the_references_mesh(name, tmp_path):
    """SURVEY s8 f4's parenthetical, round 6: "the same input etree" without the reference's mesher.  oracle/make_cvm --
    the reference's OWN etree and cvm libraries -- writes the very material database the golden run meshed; hqh_cvm_open
    reads it with this library's etree reader (B-tree pages, locational keys, the dbctl trailer of cvm_setdbctl),
    hqh_cvm_query answers as cvm_query would (cvm.c:266-311: the leaf octant that holds the point), hqh_cvm_grid hands
    the model to hqh_octree_generate in the mesh's axes (setrec queries east = y, north = x, psolve.c:1352) -- and out comes
    the reference's mesh, leaf for leaf and record for record."""
    import subprocess
    db = str(tmp_path / "model.e")
    subprocess.run([MAKE_CVM, db] + _cvm_args(name), check=True)
    cvm = host.Cvm(db)
    assert cvm.nleaves == 16 * 16 * 8 and cvm.levels == (4, 4) and cvm.region == (1000.0, 1000.0, 500.0)
    assert cvm.ticksize == 1000.0 / 2 ** 31
    vp, vs, rho, cell = cvm.grid()
    evp, evs, erho, ecell = H.cvm_grid(name)
    assert cell == ecell == 62.5 and vp.shape == (8, 16, 16)
    assert np.array_equal(vp, evp) and np.array_equal(vs, evs) and np.array_equal(rho, erho)
    # point queries as setrec makes them: (east = mesh y, north = mesh x, depth); outside the database: none
    rng = np.random.default_rng(8)
    for _ in range(200):
        x, y, z = rng.uniform(0, 1000), rng.uniform(0, 1000), rng.uniform(0, 500)
        got = cvm.query(y, x, z)
        k, j, i = int(z // 62.5), int(y // 62.5), int(x // 62.5)
        assert got == (evp[k, j, i], evs[k, j, i], erho[k, j, i])
    assert cvm.query(1000.5, 10.0, 10.0) is None and cvm.query(10.0, 10.0, 500.5) is None
    cvm.close()
    g = H.load(name)
    spec = H.CVM_MODELS[name]
    ticks, edge, edata, far, ticksize = host.octree_generate(vp, vs, rho, cell, (1000.0, 1000.0, 500.0), spec["freq"] * 8, spec["vscut"])
    et = g["elem_ticks"]
    assert len(ticks) == len(et) == int(g["total_elements"])
    assert np.array_equal(ticks.astype(np.int64), et[:, 0, :]) and np.array_equal(edge.astype(np.int64), et[:, 7, 0] - et[:, 0, 0])
    assert np.array_equal(edata[:, 1:], g["mat_vs_vp_rho"][:, [1, 0, 2]])


def test_the_references_shipped_cvm_database():
    """examples/simple/simple_case.e (tests/golden/ref_inputs: the database the reference SHIPS, written by its own tools
    years ago): 2 048 level-4 octants of one material, the region of its dbctl block."""
    cvm = host.Cvm(os.path.join(H.GOLDEN, "ref_inputs", "simple_case.e"))
    assert cvm.nleaves == 2048 and cvm.levels == (4, 4) and cvm.region == (1000.0, 1000.0, 500.0)
    vp, vs, rho, cell = cvm.grid()
    assert cell == 62.5 and vp.shape == (8, 16, 16)
    assert (vp == 6000.0).all() and (vs == 3464.0).all() and (rho == 2700.0).all()
    assert cvm.query(500.0, 500.0, 250.0) == (6000.0, 3464.0, 2700.0)
    cvm.close()


def test_octree_generate_refuses_what_it_cannot_mesh():
    vp, vs, rho, cell = H.cvm_grid("c5_basin")
    with pytest.raises(Exception):                       # the Vs rule asks for leaves below max_level
        host.octree_generate(vp, vs, rho, cell, (1000.0, 1000.0, 500.0), 40.0, 100.0, max_level=4)
    with pytest.raises(Exception):                       # the model does not cover the domain
        host.octree_generate(vp[:4], vs[:4], rho[:4], cell, (1000.0, 1000.0, 500.0), 40.0, 100.0)
    with pytest.raises(Exception):                       # 3 : 2 : 1 does not fit octor's root cube (octor.c:4130-4146)
        host.octree_generate(vp, vs, rho, cell, (1500.0, 1000.0, 500.0), 50.0, 100.0)


def test_million_element_box_has_octors_node_order():
    """tests/golden/c2_mid: the reference refined examples/simple to 128 x 128 x 64 elements; the
    fixture's sampled and loaded nodes carry the tick coordinates its mesh.e gives them.  The box
    the C host side builds numbers those nodes as octor did (octor.c:6166), and the oracle's
    uniform_mesh agrees."""
    g = H.load("c2_mid")
    nx, ny, nz = 128, 128, 64
    box = host.Box(nx, ny, nz, 1000.0 / nx, float(g["dt"]), float(g["freq"]))
    assert box.info["lenum"] == int(g["elements"]) and box.info["nharbored"] == int(g["nodes"])
    edge = int(g["edge_ticks"])

    def key(ijk):
        ijk = np.asarray(ijk, np.int64)
        return ijk[:, 0] + (nx + 1) * (ijk[:, 1] + (ny + 1) * ijk[:, 2])

    for node_ijk in (box.node_ijk, ho.uniform_mesh(nx, ny, nz)[2]):
        lut = np.full((nx + 1) * (ny + 1) * (nz + 1), -1, np.int64)
        lut[key(node_ijk)] = np.arange(len(node_ijk))
        assert np.array_equal(lut[key(g["sample_ticks"] // edge)], g["sample_lnid"])
        assert np.array_equal(lut[key(g["loaded_ticks"] // edge)], g["loaded_lnid"])
    assert np.allclose(g["edata"], (1000.0 / nx, 6000.0, 3464.0, 2700.0))
    box.close()


def test_station_lines_with_velocity_and_acceleration_columns():
    """hqh_station_header / _format_derivs / _kinematics against the reference's own station file
    (tests/golden/c1_stations_va) and the oracle's restatement of psolve.c:6705-6787."""
    g = H.load("c1_stations_va")
    lines = str(g["station0_text"]).split("\n")
    assert host.station_header(2) == lines[0] and host.station_header(0) == lines[0][:len(host.station_header(0))]
    assert ho.station_header(1) == host.station_header(1)
    for t in (0, 2, 57, 118):
        assert host.station_format(t * 1e-3, g["stations"][0, t, 1:]) == "\n" + lines[t + 1]
        assert host.station_format(t * 1e-3, g["stations"][0, t, 1:4]) == ("\n" + lines[t + 1])[:len(ho.station_line(t * 1e-3, g["stations"][0, t, 1:4]))]
    rng = np.random.default_rng(7)
    lib = host.load_library()
    import ctypes
    for derivs in (0, 1, 2):
        phi = rng.uniform(0, 0.3, 8)
        u = rng.normal(size=(3, 8, 3))
        out = np.zeros(3 * (1 + derivs))
        rc = lib.hqh_station_kinematics(phi.ctypes.data_as(ctypes.c_void_p), u[0].ctypes.data_as(ctypes.c_void_p),
                                        u[1].ctypes.data_as(ctypes.c_void_p), u[2].ctypes.data_as(ctypes.c_void_p),
                                        ctypes.c_double(1e-3), ctypes.c_int32(derivs), out.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0
        assert np.array_equal(out, ho.station_kinematics(phi, u[0], u[1], u[2], 1e-3, derivs))   # same order of operations


def test_4d_wavefield_files_equal_the_references(tmp_path):
    """hqh_wavefield_create / _write against disp.h4d and vel.h4d the reference wrote
    (tests/golden/c1_wavefield: parallel output, rate 100, 349 steps -> 4 output steps): the header
    byte for byte except the random file id and the date; the data byte for byte, from the oracle's
    loops (bit-identical to the reference's), written once as a whole and once as two partitions."""
    import ctypes
    g = H.load("c1_wavefield")
    p = H.c1_problem()
    N, E = p["N"], p["E"]
    tick = 1000.0 / 2 ** 30
    lib = host.load_library()
    fields = []
    o1, o2 = np.zeros((N, 3)), np.zeros((N, 3))
    fields.append((o2.copy(), o1.copy()))
    for k in range(3):
        ho.solver_run(p["lnid"], p["etable"], p["ntable"], o1, o2, 100 * k, 100, p["dt"],
                      loaded_lnid=g["loaded_lnid"], forces=g["forces"])
        fields.append((o2.copy(), o1.copy()))           # tm1, tm2 as the print at step 100 (k + 1) sees them
    for q, name in ((1, "disp"), (2, "vel")):
        ref = g[name + "_np1"].tobytes()
        for parts in (1, 2):
            path = str(tmp_path / ("%s%d.h4d" % (name, parts)))
            host.wavefield_create(path, {1: "displacement", 2: "velocity"}[q], N, E, (1000.0, 1000.0, 500.0), tick,
                                  p["dt"], 100, 349)
            cuts = [0, N] if parts == 1 else [0, 1000, N]
            for k, (tm1, tm2) in enumerate(fields):
                for a, b in zip(cuts[:-1], cuts[1:]):
                    # a partition holds its owned nodes somewhere in its local arrays: here at local id 5
                    l1 = np.vstack([np.full((5, 3), 7.0), tm1[a:b]])
                    l2 = np.vstack([np.full((5, 3), 9.0), tm2[a:b]])
                    rc = lib.hqh_wavefield_write(path.encode(), ctypes.c_int64(N), ctypes.c_int32(q), ctypes.c_int32(k),
                                                 ctypes.c_int64(a), ctypes.c_int32(5), ctypes.c_int32(b - a),
                                                 l1.ctypes.data_as(ctypes.c_void_p), l2.ctypes.data_as(ctypes.c_void_p),
                                                 ctypes.c_double(p["dt"]))
                    assert rc == 0
            ours = open(path, "rb").read()
            assert len(ours) == len(ref) == 136 + 4 * N * 24
            assert ours[:32] == ref[:32] and ours[48:128] == ref[48:128]     # all but ufid[16] and generation_date
            assert ours[136:] == ref[136:]
    assert np.abs(np.frombuffer(ref[136:], "<f8")).max() > 1e3


@pytest.mark.parametrize("nranks", [1, 3])
def test_laterally_varying_material_tables_equal_the_oracles_bitwise(nranks):
    """c3h's kind of mesh: hqh_box_params.lateral_classes > 1 gives every element column its own Vp, Vs, rho (a class
    factor on the layer's values), so neighbouring elements differ as on a real CVM mesh.  From the per-element edata
    the box reports, the oracle's solver_init (psolve.c:3360-3473: float-evaluated mu / lambda / zeta, dashpots, the
    nodal sums in element order) builds the same eTable and nTable bit for bit -- whole and on partitions."""
    nx, ny, nz, h, dt, freq = 16, 16, 8, 62.5, 1e-3, 5.0
    layers = [(0.0, 3000.0, 1400.0, 2200.0), (200.0, 6000.0, 3464.0, 2700.0)]
    whole = host.Box(nx, ny, nz, h, dt, freq, layers=layers, lateral_classes=61, lateral_amp=0.1)
    mat = whole.material()
    assert len(np.unique(mat[:, 1])) > 40                     # neighbouring element columns differ
    assert abs(mat[:, 1] / np.where(mat[:, 1] > 2000, 3464.0, 1400.0) - 1).max() <= 0.1 + 1e-6
    elem_ijk, lnid, node_ijk = ho.uniform_mesh(nx, ny, nz)
    assert np.array_equal(whole.lnid, lnid)
    edata = np.empty((len(lnid), 4), np.float32)
    edata[:, 0] = h
    edata[:, 1:] = mat
    et, nt = ho.solver_init(lnid, edata, ho.face_bits(elem_ijk, nx, ny, nz), len(node_ijk), dt, freq)
    assert np.array_equal(whole.etable, et)
    assert np.array_equal(whole.ntable, nt)
    gid = {tuple(v): i for i, v in enumerate(whole.node_ijk.tolist())}
    e0 = 0
    for r in range(nranks if nranks > 1 else 0):
        b = host.Box(nx, ny, nz, h, dt, freq, layers=layers, lateral_classes=61, lateral_amp=0.1, rank=r, nranks=nranks)
        E = b.info["lenum"]
        assert np.array_equal(b.etable, et[e0:e0 + E]) and np.array_equal(b.material(), mat[e0:e0 + E])
        g = np.array([gid[tuple(v)] for v in b.node_ijk.tolist()])
        assert np.array_equal(b.ntable, nt[g])               # every harbored copy is evaluated completely
        e0 += E
        b.close()
    whole.close()


def _octbox_tables(b):
    sch = b.schedules()
    return dict(lnid=b.lnid.copy(), xyz=b.node_xyz.copy(), et=b.etable.copy(), nt=b.ntable.copy(), owner=b.owner.copy(),
                dn=[np.asarray(x).copy() for x in b.dangling],
                sch={k: {lst: [(int(pr), np.asarray(m).copy()) for pr, m in sch[k][lst]] for lst in sch[k]} for k in sch})


@pytest.mark.parametrize("kind,nranks", [("two_level", 3), ("two_level", 8), ("o3s", 5), ("o3s", 8)])
def test_per_rank_construction_of_octree_boxes_equals_the_cut_of_the_whole_box(kind, nranks, monkeypatch):
    """hqh_octbox_create_levels on a partition: built from the sorted leaf keys alone (HQH_OCTBOX_LOCAL=1: what large
    boxes get -- no whole-box arrays, so that eight ranks of a 189 M-element basin do not each build all of it) against
    the whole box cut into octor's per-rank tables (octor.c:4939-4944 partition, :5466-5475 ownership, :5516-6040
    sharing): connectivity, coordinates, eTable, nTable (bit for bit: element order, then the hanging nodes' mass
    parts), owners, dnodeTable with its anchor order, both schedules."""
    import bench

    def make(rank):
        if kind == "o3s":
            return bench.make_octbox("o3s", rank, nranks)[0]
        return host.OctBox(32, 16, 6, 5, 31.25, 1e-3, 5.0, rank=rank, nranks=nranks)
    for rank in range(nranks):
        monkeypatch.setenv("HQH_OCTBOX_LOCAL", "0")
        whole = make(rank)
        a = _octbox_tables(whole)
        assert (whole.gid >= 0).all()
        whole.close()
        monkeypatch.setenv("HQH_OCTBOX_LOCAL", "1")
        loc = make(rank)
        b = _octbox_tables(loc)
        assert (loc.gid == -1).all()                          # the global node index is the one thing not computed
        loc.close()
        for k in ("lnid", "xyz", "et", "nt", "owner"):
            assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), (rank, k)
        for x, y in zip(a["dn"], b["dn"]):
            assert np.array_equal(x, y), (rank, "dangling")
        assert a["sch"].keys() == b["sch"].keys()
        for k in a["sch"]:
            for lst in a["sch"][k]:
                assert len(a["sch"][k][lst]) == len(b["sch"][k][lst]), (rank, k, lst)
                for (p1, m1), (p2, m2) in zip(a["sch"][k][lst], b["sch"][k][lst]):
                    assert p1 == p2 and np.array_equal(m1, m2), (rank, k, lst, p1)
