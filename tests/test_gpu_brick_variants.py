"""The brick kernels that no mesh of solver_init reaches by itself, against the oracle (through the C-ABI):
hq_k_brick<true> -- uniform element coefficients, n_t rows that differ from node to node -- and the planner
switches around the bricks (HQ_BRICK_CZ / MINZ / MINNODES / NO_HET) on a box small enough for the oracle's whole run.
Tolerance as everywhere: <= 1e-9 relative L-inf on nodal displacement (psolve.c:4072-4114 + stiffness.c:180-237 +
damping.c:29-103 in fp64, other summation order)."""
import numpy as np
import pytest

import hercules_amd as ha
from oracle import herc_oracle as ho
from tests import helpers as H

pytestmark = pytest.mark.gpu

TOL = 1e-9


def _ticks(node_ijk, edge=1 << 24):
    return (np.asarray(node_ijk, np.int64) * edge).astype(np.int32)


def _box(nx, ny, nz, h=15.0, dt=3e-4, freq=30.0):
    elem_ijk, lnid, node_ijk = ho.uniform_mesh(nx, ny, nz)
    edata = np.empty((len(lnid), 4), np.float32)
    edata[:] = (h, 6000.0, 3464.0, 2700.0)
    face = ho.face_bits(elem_ijk, nx, ny, nz)
    et, nt = ho.solver_init(lnid, edata, face, len(node_ijk), dt, freq)
    return lnid, node_ijk, et, nt, dt


def _run_both(lnid, node_ijk, et, nt, dt, nsteps, seed=7):
    N = len(node_ijk)
    rng = np.random.default_rng(seed)
    u1 = rng.uniform(-1, 1, (N, 3)) * 1e-3
    u2 = u1 + rng.uniform(-1, 1, (N, 3)) * 1e-6
    s = ha.Solver(lnid, et, nt, dt, node_xyz=_ticks(node_ijk), tm1=u1, tm2=u2, variant=ha.HQ_VARIANT_PATCH)
    info = s.info()
    s.run(nsteps)
    tm1, tm2 = s.download()
    s.close()
    o1, o2 = u2.copy(), u1.copy()
    ho.solver_run(lnid, et.copy(), nt.copy(), o1, o2, 0, nsteps, dt)
    return info, tm1, tm2, o2, o1


def test_brick_units_with_per_node_nt_rows_forced(monkeypatch):
    """HQ_BRICK_NO_NTSAME=1: every uniform unit reads its nodes' own n_t rows -- hq_k_brick<true> on the 64 x 64 x 32 box."""
    monkeypatch.setenv("HQ_BRICK_NO_NTSAME", "1")
    lnid, node_ijk, et, nt, dt = _box(64, 64, 32)
    info, tm1, tm2, r1, r2 = _run_both(lnid, node_ijk, et, nt, dt, 12)
    assert info["brick_units"] > 0 and info["brick_units_pernode"] == info["brick_units"] and info["brick_units_het"] == 0
    assert H.rel_linf(tm1, r1) < TOL and H.rel_linf(tm2, r2) < TOL


def test_brick_units_with_a_callers_per_node_masses():
    """A caller's nTable whose rows differ from node to node inside a homogeneous region (mass_simple, mass2_minusaM and
    mass_minusaM each scaled by a factor of the node's own; the three axes equal, so the nodes stay dashpot-free): the
    planner must give those units per-node rows, and hq_k_brick<true> must reproduce solver_compute_displacement
    (psolve.c:4078-4106) with them."""
    lnid, node_ijk, et, nt, dt = _box(64, 64, 32)
    rng = np.random.default_rng(99)
    ijk = np.asarray(node_ijk)
    inside = np.all((ijk > 0) & (ijk < np.array([64, 64, 32])), axis=1)
    nt = nt.copy()
    nt[inside, 0] *= rng.uniform(0.8, 1.25, inside.sum())
    nt[inside, 1:4] *= rng.uniform(0.9, 1.1, inside.sum())[:, None]
    nt[inside, 4:7] *= rng.uniform(0.9, 1.1, inside.sum())[:, None]
    info, tm1, tm2, r1, r2 = _run_both(lnid, node_ijk, et, nt, dt, 12)
    assert info["brick_units_pernode"] > 0 and info["brick_units_pernode"] == info["brick_units"]
    assert H.rel_linf(tm1, r1) < TOL and H.rel_linf(tm2, r2) < TOL


@pytest.mark.parametrize("env", [{"HQ_BRICK_CZ": "5"}, {"HQ_BRICK_CZ": "64", "HQ_BRICK_MINZ": "9"},
                                 {"HQ_BRICK_MINNODES": "100000"}, {"HQ_BRICK_NO_HET": "1"}])
def test_brick_planner_switches(monkeypatch, env):
    """The planner's switches that no other GPU test sets: odd unit lengths, long minimum runs, a node threshold that
    leaves everything to the patches, and no HET units on a box with two materials side by side (the nodes of the
    material interface then stay with the patches)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    nx, ny, nz = 64, 32, 24
    elem_ijk, lnid, node_ijk = ho.uniform_mesh(nx, ny, nz)
    edata = np.empty((len(lnid), 4), np.float32)
    edata[:] = (15.0, 6000.0, 3464.0, 2700.0)
    # a second material beside the first (a vertical interface: the tiles that straddle it become HET units) or, without
    # HET units, below it (a horizontal interface cuts the tile columns' runs; a vertical one would leave no full tile)
    axis, cut = (2, 14) if "HQ_BRICK_NO_HET" in env else (0, 40)
    edata[np.asarray(elem_ijk)[:, axis] >= cut] = (15.0, 5000.0, 2800.0, 2500.0)
    face = ho.face_bits(elem_ijk, nx, ny, nz)
    dt = 3e-4
    et, nt = ho.solver_init(lnid, edata, face, len(node_ijk), dt, 30.0)
    info, tm1, tm2, r1, r2 = _run_both(lnid, node_ijk, et, nt, dt, 10)
    if "HQ_BRICK_MINNODES" in env:
        assert info["brick_units"] == 0
    else:
        assert info["brick_units"] > 0
    if "HQ_BRICK_NO_HET" in env:
        assert info["brick_units_het"] == 0
    assert H.rel_linf(tm1, r1) < TOL and H.rel_linf(tm2, r2) < TOL


@pytest.mark.parametrize("env", [{"HQ_PATCH_PSPLIT": "128", "HQ_PATCH_PMERGE": "128"}, {"HQ_PATCH_THREADS": "256"},
                                 {"HQ_PATCH_NLMAX": "640", "HQ_PATCH_PMAX": "256"}, {"HQ_PATCH_PIPE": "0"}])
def test_patch_planner_switches(monkeypatch, env):
    """The patch planner's tuning switches that no other GPU test sets (profiles/sweep_patch_cfg.sh uses them): smaller
    patches, 256-thread workgroups, a smaller LDS image, the one-patch kernel forced -- with and without bricks -- on a
    two-level octree box (hanging nodes: element-form patches with extra accumulators) against the oracle."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ref = H.two_level_mesh(32, 16, 6, 5)
    rng = np.random.default_rng(11)
    u1 = rng.uniform(-1, 1, (ref["N"], 3)) * 1e-3
    u2 = u1 + rng.uniform(-1, 1, (ref["N"], 3)) * 1e-6
    ho.compute_adjust(u1, 1, ref["dangling"])
    ho.compute_adjust(u2, 1, ref["dangling"])
    nsteps = 8
    o1, o2 = u2.copy(), u1.copy()
    ho.solver_run(ref["lnid"], ref["etable"].copy(), ref["ntable"].copy(), o1, o2, 0, nsteps, ref["dt"], dangling=ref["dangling"])
    for nobricks in ("0", "1"):
        monkeypatch.setenv("HQ_NO_BRICKS", nobricks)
        s = ha.Solver(ref["lnid"], ref["etable"], ref["ntable"], ref["dt"], dangling=ref["dangling"],
                      node_xyz=ref["node_q"], tm1=u1, tm2=u2, variant=ha.HQ_VARIANT_PATCH)
        s.run(nsteps)
        tm1, tm2 = s.download()
        s.close()
        assert H.rel_linf(tm1, o2) < TOL and H.rel_linf(tm2, o1) < TOL


def test_a_group_with_a_destroyed_member_refuses_to_step():
    """hq_group_run on a group one of whose members was destroyed: HQ_ERR_STATE, not stores into freed buffers."""
    from hercules_amd import capi, host
    boxes = [host.Box(32, 32, 16, 15.0, 3e-4, 30.0, rank=r, nranks=2) for r in range(2)]
    solvers = [b.create_solver() for b in boxes]
    capi.group_link(solvers)
    capi.group_run(solvers, 2)
    solvers[1].close()
    with pytest.raises(ha.HqError):
        capi.group_run(solvers[:1], 1)
    solvers[0].close()
    for b in boxes:
        b.close()


@pytest.mark.parametrize("pernode", [False, True])
def test_brick_kernel_in_the_form_partitions_launch(monkeypatch, pernode):
    """hq_k_brick<PERNODE, BYCOMP = true>: the plane sums component by component (100 / 108 VGPRs: room for the exchange
    chain beside it) -- what every context with a transport launches -- forced on a single context against the oracle."""
    monkeypatch.setenv("HQ_BRICK_BY_COMPONENT", "1")
    if pernode:
        monkeypatch.setenv("HQ_BRICK_NO_NTSAME", "1")
    lnid, node_ijk, et, nt, dt = _box(64, 64, 32)
    info, tm1, tm2, r1, r2 = _run_both(lnid, node_ijk, et, nt, dt, 12, seed=3)
    assert info["brick_units"] > 0 and (info["brick_units_pernode"] == info["brick_units"]) == pernode
    assert H.rel_linf(tm1, r1) < TOL and H.rel_linf(tm2, r2) < TOL


def test_typed_options_select_kernels_per_context(monkeypatch):
    """hq_options (ABI 6) instead of the environment: two contexts of ONE process differ -- one with bricks, one with
    hq_options.no_bricks = 1 and hq_k_patch_pers (patch_pipe = 4) -- both against the oracle; hq_get_options returns what
    each runs with; an HQ_* variable in the environment overrides a field only where the caller allows it."""
    lnid, node_ijk, et, nt, dt = _box(96, 40, 24)
    rng = np.random.default_rng(11)
    u1 = rng.uniform(-1, 1, (len(nt), 3)) * 1e-3
    u2 = u1 * 0.999
    nsteps = 5
    o1, o2 = u2.copy(), u1.copy()
    ho.solver_run(lnid, et, nt, o1, o2, 0, nsteps, dt)
    a = ha.Solver(lnid, et, nt, dt, tm1=u1, tm2=u2, node_xyz=_ticks(node_ijk), variant=ha.HQ_VARIANT_PATCH)
    b = ha.Solver(lnid, et, nt, dt, tm1=u1, tm2=u2, node_xyz=_ticks(node_ijk), variant=ha.HQ_VARIANT_PATCH,
                  options={"no_bricks": 1, "patch_pipe": 4})
    c = ha.Solver(lnid, et, nt, dt, tm1=u1, tm2=u2, node_xyz=_ticks(node_ijk), variant=ha.HQ_VARIANT_PATCH,
                  options=ha.capi.Options(brick_cz=6, brick_by_component=1))
    a_nodes = a.info()["brick_nodes"]
    assert a.info()["brick_nodes"] > 0 and b.info()["brick_nodes"] == 0
    assert c.info()["brick_units"] > a.info()["brick_units"]
    oa, ob, oc = a.options(), b.options(), c.options()
    assert oa["no_bricks"] == -1 and ob["no_bricks"] == 1 and ob["patch_pipe"] == 4 and oc["brick_cz"] == 6
    for s in (a, b, c):
        s.run(nsteps)
        tm1, tm2 = s.download()
        assert H.rel_linf(tm1, o2) < 1e-9 and H.rel_linf(tm2, o1) < 1e-9
        s.close()
    # The environment (ABI 6): an HQ_* variable overrides a field only where the caller allows it -- allow_env = 1, or the
    # default in a process that says HQ_ALLOW_ENV=1 (this suite: tests/conftest.py); a host that passes allow_env = 0
    # (examples/psolve_hq_stub.inc) is not steered by it.  Resolved once: hq_get_options returns what really runs.
    mk = lambda **o: ha.Solver(lnid, et, nt, dt, tm1=u1, tm2=u2, node_xyz=_ticks(node_ijk), variant=ha.HQ_VARIANT_PATCH, options=o)
    monkeypatch.setenv("HQ_NO_BRICKS", "1")
    d = mk(no_bricks=0)
    assert d.info()["brick_nodes"] == 0 and d.options()["no_bricks"] == 1 and d.options()["allow_env"] == 1
    d.close()
    d = mk(no_bricks=0, allow_env=0)
    assert d.info()["brick_nodes"] > 0 and d.options()["no_bricks"] == 0 and d.options()["allow_env"] == 0
    d.close()
    monkeypatch.delenv("HQ_ALLOW_ENV")                       # a process that does not opt in: the variable is ignored
    d = mk()
    assert d.info()["brick_nodes"] > 0 and d.options()["no_bricks"] == -1 and d.options()["allow_env"] == 0
    d.close()
    d = mk(allow_env=1)
    assert d.info()["brick_nodes"] == 0
    d.close()
    monkeypatch.setenv("HQ_ALLOW_ENV", "1")
    monkeypatch.setenv("HQ_NO_BRICKS", "0")                  # a switch set to 0 is OFF, and reported as 0 (round-5 advisor)
    monkeypatch.setenv("HQ_BRICK_NO_FACES", "0")
    d = mk()
    assert d.info()["brick_nodes"] == a_nodes and d.options()["no_bricks"] == 0 and d.options()["brick_no_faces"] == 0
    d.close()
    monkeypatch.setenv("HQ_BRICK_NO_FACES", "1")
    d = mk()
    assert 0 < d.info()["brick_nodes"] < a_nodes and d.options()["brick_no_faces"] == 1
    d.close()


def test_no_ntsame_through_the_typed_options_reaches_every_unit():
    """hq_options.brick_no_ntsame = 1 WITHOUT the environment (round-5 advisor, medium: the option was read inside an
    OpenMP region, whose worker threads saw no options -- only the units filled by the calling thread obeyed): every
    uniform unit must come out with n_t rows of its own, on a box large enough for the planner's loop to be shared
    between threads; against the oracle."""
    lnid, node_ijk, et, nt, dt = _box(128, 64, 32)
    N = len(node_ijk)
    rng = np.random.default_rng(5)
    u1 = rng.uniform(-1, 1, (N, 3)) * 1e-3
    u2 = u1 * 0.999
    s = ha.Solver(lnid, et, nt, dt, node_xyz=_ticks(node_ijk, 1 << 22), tm1=u1, tm2=u2, variant=ha.HQ_VARIANT_PATCH,
                  options={"brick_no_ntsame": 1, "allow_env": 0})
    info = s.info()
    assert info["brick_units"] >= 16 and info["brick_units_pernode"] == info["brick_units"] and info["brick_units_het"] == 0
    s.run(6)
    tm1, tm2 = s.download()
    s.close()
    o1, o2 = u2.copy(), u1.copy()
    ho.solver_run(lnid, et.copy(), nt.copy(), o1, o2, 0, 6, dt)
    assert H.rel_linf(tm1, o2) < TOL and H.rel_linf(tm2, o1) < TOL


def test_two_materials_in_one_tile_footprint_against_the_oracle():
    """tests/helpers.two_material_leaves: a material boundary inside one level, off the tile grid -- the footprints that
    straddle it carry two ragged columns (HQ_BK_RAGGED), one per material; the boundary plane itself is the patches'.
    Three steps from a seeded field against the oracle's reference loops, with and without ragged columns."""
    from hercules_amd import host
    ticks, edge, edata, far = H.two_material_leaves()
    box = host.OctBox.from_leaves(ticks, edge, edata, far, 1e-3, 2.0)
    xyz = box.node_xyz.astype(np.int64)
    gid = (xyz[:, 2] * 1000003 + xyz[:, 1]) * 1000003 + xyz[:, 0]
    u = np.empty((box.N, 3))
    for d in range(3):
        x = (gid * 3 + d + 7) * np.int64(2654435761) % np.int64(2 ** 31)
        u[:, d] = (x.astype(np.float64) / 2 ** 30 - 1.0) * 1e-3
    nsteps = 3
    o1, o2 = (0.999 * u).copy(), u.copy()
    ho.solver_run(box.lnid, box.etable.copy(), box.ntable.copy(), o1, o2, 0, nsteps, box.dt)
    for ragged in (1, 0):
        s = box.create_solver(variant=ha.HQ_VARIANT_PATCH, tm1=u, tm2=0.999 * u, options={"brick_ragged": ragged})
        info = s.info()
        # (without ragged columns the 32-wide round and then the per-element units take what they can of the straddling tiles)
        assert (info["brick_units_ragged"] >= 8) == bool(ragged) and (info["brick_units_het"] == 0) == bool(ragged)
        s.run(nsteps)
        tm1, tm2 = s.download()
        s.close()
        assert H.rel_linf(tm1, o2) < 1e-9 and H.rel_linf(tm2, o1) < 1e-9
    box.close()
