"""The patch planner of libhq_solver.so is host code: hq_plan_check plans a mesh exactly as hq_create
would and checks the plan against the mesh WITHOUT a device (no compute, nothing stepped):
every element row names the LDS rows of its element's eight nodes, accumulate flags = owned nodes +
hanging nodes on owned anchors, every node owned by exactly one patch.  It also counts the LDS passes
of the element gathers under the bank rule of MI355X_MICROARCH.md (32-lane groups, 24-byte rows
conflict iff equal modulo 32): lattice patches must be free of conflicts."""
import os

import numpy as np
import pytest

from hercules_amd import host


def test_uniform_box_interior_patches_are_conflict_free_lattices():
    b = host.Box(64, 64, 32, 10.0, 2e-4, 50.0)
    r = b.plan_check()
    b.close()
    assert r["faults"] == 0
    assert r["patches"] == 8 * 8 * 4
    assert r["lattice_patches"] == 6 * 6 * 2                  # the patches with a full ring of neighbours
    # 729 elements = 23 groups of 32 lanes, 8 corners: one pass each
    assert r["lattice_gather_passes"] == r["lattice_patches"] * 23 * 8
    # all lattice patches share ONE element-row block
    assert r["distinct_row_blocks"] <= r["patches"] - r["lattice_patches"] + 1
    assert r["gather_passes"] < 2.0 * r["gather_instructions"]


def test_lattice_can_be_switched_off(monkeypatch):
    monkeypatch.setenv("HQ_PATCH_NO_LATTICE", "1")
    b = host.Box(32, 32, 32, 10.0, 2e-4, 50.0)
    r = b.plan_check()
    b.close()
    assert r["faults"] == 0 and r["lattice_patches"] == 0


@pytest.mark.parametrize("nranks", [1, 4])
def test_octree_box_plan(nranks):
    """Four octree levels with hanging nodes (bench.py's o3s), whole and cut into partitions: patches
    around hanging nodes keep the id-ordered rows, uniform regions of every level become lattices."""
    import bench
    tot = {"patches": 0, "lattice_patches": 0, "pairs": 0}
    for r in range(nranks):
        box, E, N, _ = bench.make_octbox("o3s", r, nranks)
        rep = box.plan_check()
        box.close()
        assert rep["faults"] == 0
        if rep["lattice_patches"]:
            assert rep["lattice_gather_passes"] == rep["lattice_patches"] * 23 * 8
        for k in tot:
            tot[k] += rep[k]
    assert tot["lattice_patches"] > 0.2 * tot["patches"]
    assert tot["pairs"] >= E


def test_stencil_tables_match_the_mesh_on_a_uniform_box():
    """hq_stencil_plan_check: every patch of a uniform box is a lattice subset -- the interior ones full lattices,
    the others ragged (ring missing at a domain face, 9-wide far-face cubes); their shape tables (rows, element masks,
    boundary lists) agree with the connectivity, and the element-matrix blocks of the boundary phase reproduce the
    kernels' element arithmetic for every kind of present-octant set."""
    b = host.Box(64, 64, 32, 10.0, 2e-4, 50.0)
    r = b.stencil_plan_check()
    b.close()
    assert r["faults"] == 0
    assert r["patches"] == 8 * 8 * 4 and r["tables"] == r["patches"]
    assert r["full_lattices"] == 6 * 6 * 2
    # boundary nodes: the owned nodes on the six domain faces (65 x 65 x 33 nodes)
    assert r["boundary_nodes"] == 65 * 65 * 33 - 63 * 63 * 31
    assert r["corners_checked"] > 8 * 64 * 64 * 32


@pytest.mark.parametrize("nranks", [1, 4])
def test_stencil_tables_on_an_octree_box(nranks):
    """The four-level octree box (hanging nodes), whole and as partitions: patches with hanging-node work keep the
    element form; every table that IS made agrees with the mesh."""
    import bench
    tables = patches = 0
    for rank in range(nranks):
        box = bench.make_octbox("o3s", rank, nranks)[0]
        rep = box.stencil_plan_check()
        box.close()
        assert rep["faults"] == 0
        tables += rep["tables"]
        patches += rep["patches"]
    assert 0 < tables < patches


def test_stencil_tables_on_the_partitions_of_a_uniform_box():
    """Two block partitions of a 64^3 box: every cube patch is a lattice subset with a table (the patches on the
    partition interface among them: ragged); only the patches made of the interface PLANE's nodes alone -- rank 0
    harbors them behind its cubes, octor numbers them with the next partition's cells -- are no lattice subsets and
    keep the element form."""
    for rank, plane_patches in ((0, 16), (1, 0)):
        b = host.Box(64, 64, 64, 10.0, 2e-4, 50.0, rank=rank, nranks=2)
        r = b.stencil_plan_check()
        b.close()
        assert r["faults"] == 0
        assert r["tables"] == 256 and r["patches"] == 256 + plane_patches
        assert 0 < r["full_lattices"] < r["tables"]


def test_brick_plan_of_a_uniform_box_matches_the_connectivity():
    """hq_brick_plan_check: the simple nodes of a uniform box -- everything but the six faces -- become tile columns of
    64 x 8 nodes; every neighbour hq_k_brick will read (unit-internal, ring table, first / last plane lists) is the node
    the connectivity says, the numbering is a permutation, the patches keep the rest.  Since round 5 the interior of the
    two faces normal to z rides with the columns under / above it (the free surface and the bottom dashpot face: first /
    last plane of the first / last unit, 17 neighbours each); HQ_BRICK_NO_FACES=1 leaves them to the patches."""
    b = host.Box(64, 64, 32, 10.0, 2e-4, 50.0)
    r = b.brick_plan_check()
    os.environ["HQ_BRICK_NO_FACES"] = "1"
    try:
        r0 = b.brick_plan_check()
    finally:
        del os.environ["HQ_BRICK_NO_FACES"]
    b.close()
    assert r["faults"] == 0 and r0["faults"] == 0
    assert r0["brick_nodes"] == 63 * 63 * 31 and r0["patch_nodes"] == 65 * 65 * 33 - 63 * 63 * 31
    assert r0["neighbours_checked"] == 26 * r0["brick_nodes"]
    assert r["brick_nodes"] == 63 * 63 * 33 and r["patch_nodes"] == 65 * 65 * 33 - 63 * 63 * 33
    assert r["neighbours_checked"] == 26 * 63 * 63 * 31 + 17 * 63 * 63 * 2
    for q in (r, r0):
        assert q["columns"] == 8 and q["units"] == 8 * 4 and q["units_one_nt_row"] == q["units"] and q["het_units"] == 0   # 31 planes in chunks of 8: a small mesh


def test_brick_plan_on_partitions_layers_and_lateral_material():
    """Partitions (interface nodes stay with the patches), a layered box (the units of each layer carry its coefficients
    and its n_t row; the nodes of the interface plane have elements of two materials around them: one plane is too
    short a run, they stay with the patches) and a box whose material differs from element to element (tiles of
    63 x 7 nodes for hq_k_brick_het, every unit with its own element coefficients)."""
    tot = 0
    for rank in range(2):
        b = host.Box(64, 64, 64, 10.0, 2e-4, 50.0, rank=rank, nranks=2)
        r = b.brick_plan_check()
        b.close()
        assert r["faults"] == 0 and r["brick_nodes"] > 0
        tot += r["brick_nodes"]
    assert tot == 63 * 63 * (31 + 31 + 2)                    # the interface plane z = 32 and the x / y faces are left out;
                                                             # the free surface rides with rank 0's columns, the bottom face with rank 1's
    layers = [(0.0, 3000.0, 1400.0, 2200.0), (200.0, 6000.0, 3464.0, 2700.0)]
    b = host.Box(32, 32, 32, 12.5, 2e-4, 50.0, layers=layers)
    r = b.brick_plan_check()
    b.close()
    assert r["faults"] == 0 and r["brick_nodes"] == 31 * 31 * (15 + 15 + 2) and r["units_one_nt_row"] == r["units"]
    b = host.Box(32, 32, 32, 12.5, 2e-4, 50.0, lateral_classes=61, lateral_amp=0.1)
    r = b.brick_plan_check()
    assert r["faults"] == 0 and r["brick_nodes"] == 31 ** 3 and r["columns"] == 5 and r["het_units"] == r["units"] == 5 * 4 and r["units_one_nt_row"] == 0
    os.environ["HQ_BRICK_NO_HET"] = "1"                       # the uniform-coefficient kernel alone finds nothing here
    try:
        r = b.brick_plan_check()
    finally:
        del os.environ["HQ_BRICK_NO_HET"]
    b.close()
    assert r["faults"] == 0 and r["brick_nodes"] == 0 and r["patch_nodes"] == 33 ** 3


@pytest.mark.parametrize("nranks", [1, 4])
def test_brick_plan_on_an_octree_box(nranks):
    """The four-level octree box (hanging nodes), whole and as partitions: every level's uniform interior is planned
    on its own lattice; hanging nodes, anchors and interface nodes stay with the patches."""
    import bench
    bricks = nodes = 0
    for rank in range(nranks):
        box = bench.make_octbox("o3s", rank, nranks)[0]
        rep = box.brick_plan_check()
        box.close()
        assert rep["faults"] == 0
        bricks += rep["brick_nodes"]
        nodes += rep["brick_nodes"] + rep["patch_nodes"]
    assert bricks > 0.3 * nodes


@pytest.mark.parametrize("nranks", [1, 3])
def test_ragged_brick_units_beside_lateral_level_interfaces(nranks, monkeypatch):
    """The laterally refined basin o4s (level interfaces with x-, y- and z-normal faces along a sediment bowl): few 64 x 8
    tiles are FULL of uniform simple nodes over a run of planes; the second planner round takes partly filled ones
    (HQ_BK_RAGGED: every plane through an id table, owned nodes numbered without gaps, all of one material and n_t row).
    hq_brick_plan_check follows every owned node's 26 neighbours through those tables; the patches shrink to less than half."""
    import bench
    tot = {}
    for ragged in (0, 1):
        monkeypatch.setenv("HQ_BRICK_RAGGED", str(ragged))
        acc = {"brick_nodes": 0, "patch_nodes": 0, "ragged_units": 0, "ragged_nodes": 0}
        for rank in range(nranks):
            box = bench.make_octbox("o4s", rank, nranks)[0]
            rep = box.brick_plan_check()
            box.close()
            assert rep["faults"] == 0 and rep["units_one_nt_row"] == rep["units"] - rep["het_units"]
            for k in acc:
                acc[k] += rep[k]
        tot[ragged] = acc
    assert tot[0]["ragged_units"] == 0 and tot[0]["ragged_nodes"] == 0
    assert tot[1]["ragged_units"] > 0 and tot[1]["ragged_nodes"] > 0.15 * (tot[1]["brick_nodes"] + tot[1]["patch_nodes"])
    assert tot[1]["patch_nodes"] < 0.6 * tot[0]["patch_nodes"]
    assert tot[1]["brick_nodes"] + tot[1]["patch_nodes"] == tot[0]["brick_nodes"] + tot[0]["patch_nodes"]


def test_ragged_columns_of_two_materials_share_a_footprint():
    """A material boundary inside one level, off the tile grid (element column 37 of a 128 x 32 x 32 box): the 64 x 8
    footprints that straddle it hold simple nodes of TWO materials -- two ragged columns per footprint, one material and one
    n_t row each (the planner's second and later passes over a tile); the nodes ON the boundary plane stay with the patches.
    hq_brick_plan_check compares every owned node's eight elements with its unit's coefficients."""
    from tests import helpers as H
    ticks, edge, edata, far = H.two_material_leaves()
    box = host.OctBox.from_leaves(ticks, edge, edata, far, 1e-3, 2.0)
    rep = box.brick_plan_check()
    box.close()
    assert rep["faults"] == 0 and rep["het_units"] == 0
    # x = 1 .. 36 of the one material and 38 .. 64 of the other, in every one of the 31 x 31 interior rows and planes
    assert rep["ragged_nodes"] == (36 + 27) * 31 * 31 and rep["ragged_units"] >= 8
    # ... beside the full tiles of x = 65 .. 127 (with the z faces of their columns: 33 planes)
    assert rep["brick_nodes"] == rep["ragged_nodes"] + 63 * 31 * 33


def test_ragged_columns_on_the_references_own_lateral_mesh(monkeypatch):
    """tests/golden/c5_basin (meshed by the real reference: three levels, hanging nodes of all six kinds) with the planner's
    thresholds lowered until this small mesh carries ragged tile columns beside its level interfaces: every owned node's 26
    neighbours through the id tables, none of them a hanging node or an anchor."""
    from hercules_amd import capi
    from tests import helpers as H
    p = H.c5_problem("c5_basin")
    for k, v in (("HQ_BRICK_RAGGED_MINFILL", "12"), ("HQ_BRICK_MINNODES", "48"), ("HQ_BRICK_MINZ", "2")):
        monkeypatch.setenv(k, v)
    lnid, et, nt = [np.ascontiguousarray(a) for a in (p["lnid"].astype(np.int32), p["etable"], p["ntable"])]
    xyz = (p["node_q"].astype(np.int64) * p["emin"]).astype(np.int32)
    ids, ptr, anc = [np.ascontiguousarray(x, np.int32) for x in p["dangling"]]
    d = capi._Desc()
    d.lenum, d.nharbored, d.ldnnum = len(lnid), len(nt), len(ids)
    d.lnid, d.eTable, d.nTable, d.node_xyz = capi._ptr(lnid), capi._ptr(et), capi._ptr(nt), capi._ptr(xyz)
    d.dn_ldnid, d.dn_ptr, d.dn_lanid = capi._ptr(ids), capi._ptr(ptr), capi._ptr(anc)
    d.deltaT, d.rank, d.nranks = 1e-3, 0, 1
    rep = capi.brick_plan_check(d)
    assert rep["faults"] == 0 and rep["ragged_units"] >= 2 and rep["ragged_nodes"] > 500
    assert rep["brick_nodes"] + rep["patch_nodes"] == p["N"]


def test_ragged_per_element_columns_on_the_references_gradient_mesh(monkeypatch):
    """tests/golden/c5_gradient (the reference's laterally refined basin with a material of its own in every database
    octant) with the planner's thresholds lowered until this small mesh carries RAGGED units of the per-element kernel
    (hq_k_brick_het<., RAGGED>, round 6): no two neighbouring coarse elements share (c1, c2, beta), so the one-material
    ragged columns find (next to) nothing and the nodes beside the level interfaces would stay with the patches.  Every
    owned node's 26 neighbours through the id tables, its eight elements' coefficients in the unit's block."""
    from hercules_amd import capi
    from tests import helpers as H
    p = H.c5_problem("c5_gradient")
    for k, v in (("HQ_BRICK_RAGGED_MINFILL", "12"), ("HQ_BRICK_MINNODES", "48"), ("HQ_BRICK_MINZ", "2")):
        monkeypatch.setenv(k, v)
    lnid, et, nt = [np.ascontiguousarray(a) for a in (p["lnid"].astype(np.int32), p["etable"], p["ntable"])]
    xyz = (p["node_q"].astype(np.int64) * p["emin"]).astype(np.int32)
    ids, ptr, anc = [np.ascontiguousarray(x, np.int32) for x in p["dangling"]]
    d = capi._Desc()
    d.lenum, d.nharbored, d.ldnnum = len(lnid), len(nt), len(ids)
    d.lnid, d.eTable, d.nTable, d.node_xyz = capi._ptr(lnid), capi._ptr(et), capi._ptr(nt), capi._ptr(xyz)
    d.dn_ldnid, d.dn_ptr, d.dn_lanid = capi._ptr(ids), capi._ptr(ptr), capi._ptr(anc)
    d.deltaT, d.rank, d.nranks = 1e-3, 0, 1
    rep = capi.brick_plan_check(d)
    assert rep["faults"] == 0 and rep["ragged_het_units"] >= 2 and rep["ragged_het_nodes"] > 500
    assert rep["brick_nodes"] + rep["patch_nodes"] == p["N"]
    monkeypatch.setenv("HQ_BRICK_RAGGED_HET", "0")
    off = capi.brick_plan_check(d)
    assert off["faults"] == 0 and off["ragged_het_units"] == 0 and off["brick_nodes"] < rep["brick_nodes"]


def test_ragged_per_element_columns_of_the_small_gradient_basin():
    """bench.py's o4gs (0.93 M elements on four levels, 6 257 distinct materials): the full 62 x 7 tiles of the per-element
    kernel take 0.40 M of its 0.98 M nodes, the ragged ones 0.42 M more -- what is left to the patches drops from 0.58 M
    to 0.16 M nodes."""
    import bench
    box, E, N, it = bench.make_octbox("o4gs", 0, 1)
    rep = box.brick_plan_check()
    box.close()
    assert rep["faults"] == 0 and rep["units_one_nt_row"] == 0
    assert rep["ragged_het_units"] > 100 and rep["ragged_het_nodes"] > 350000 and rep["patch_nodes"] < 0.2 * N
