"""Parity of the HIP path (through the C-ABI) against the oracle and the
reference's own golden checkpoints.  Tolerance: <= 1e-9 relative L-inf on nodal
displacement (fp64; the GPU sums element forces in a different order), the bar
SURVEY.md s8c sets."""
import numpy as np
import pytest

import hercules_amd as ha
from oracle import herc_oracle as ho
from tests import helpers as H

pytestmark = pytest.mark.gpu

TOL = 1e-9


@pytest.fixture(params=["bricks", "patches-only"], autouse=True)
def brick_mode(request, monkeypatch):
    """Every parity case twice: as shipped -- the simple nodes of uniform regions stepped by hq_k_brick on the
    device's own tile-major numbering, the patches taking the rest -- and with HQ_NO_BRICKS=1, where the patch
    kernels (stencil and element form) take every node, as on meshes without node coordinates."""
    if request.param == "patches-only":
        monkeypatch.setenv("HQ_NO_BRICKS", "1")
    else:
        monkeypatch.delenv("HQ_NO_BRICKS", raising=False)
    return request.param

VARIANTS = [ha.HQ_VARIANT_SCATTER, ha.HQ_VARIANT_PATCH]


def _ticks(node_ijk, edge=1 << 26):
    return (np.asarray(node_ijk, np.int64) * edge).astype(np.int32)


def _box(nx, ny, nz, h=62.5, dt=1e-3, freq=5.0, damping=ho.DAMP_RAYLEIGH, vp=6000.0, vs=3464.0, rho=2700.0):
    elem_ijk, lnid, node_ijk = ho.uniform_mesh(nx, ny, nz)
    edata = np.empty((len(lnid), 4), np.float32)
    edata[:] = (h, vp, vs, rho)
    face = ho.face_bits(elem_ijk, nx, ny, nz)
    et, nt = ho.solver_init(lnid, edata, face, len(node_ijk), dt, freq, damping=damping)
    return lnid, node_ijk, elem_ijk, et, nt


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("case,damping", [("c1_short", "rayleigh"), ("c1_none", "none"), ("c1_mass", "mass"),
                                          ("c1_conv", "rayleigh")])
def test_c1_against_reference_checkpoints(variant, case, damping):
    """examples/simple driven by the reference's own force file; compared with the
    checkpoints the REAL reference wrote at steps 400 and 800.  c1_conv: the reference run with
    stiffness_calculation_method = conventional (compute_addforce_conventional, stiffness.c:121-174):
    the same operator K u summed in another order, so the one fused HIP formulation must match it too."""
    g = H.load(case)
    p = H.c1_problem(damping)
    s = ha.Solver(p["lnid"], p["etable"], p["ntable"], p["dt"], node_xyz=_ticks(p["node_ijk"]), variant=variant)
    assert s.info()["variant"] == variant
    s.set_source(g["loaded_lnid"], g["forces"])
    done = 0
    for k, step in enumerate(g["ckpt_steps"]):
        s.run(int(step) - done)
        done = int(step)
        tm1, tm2 = s.download()
        assert H.rel_linf(tm1, g["ckpt_tm1"][k]) < TOL
        assert H.rel_linf(tm2, g["ckpt_tm2"][k]) < TOL
    s.close()


@pytest.mark.parametrize("variant", VARIANTS)
def test_c1_full_run_stations(variant):
    """20 000 steps; stations gathered every step window; against the traces the
    reference ships (printed to 7 digits) and its checkpoints at 12 000 / 18 000."""
    g = H.load("c1_full")
    p = H.c1_problem()
    ids, phi = ho.station_weights(H.C1_STATIONS, H.C1_H, H.C1_NX, H.C1_NY, H.C1_NZ, p["lnid"], p["elem_ijk"])
    s = ha.Solver(p["lnid"], p["etable"], p["ntable"], p["dt"], node_xyz=_ticks(p["node_ijk"]), variant=variant)
    s.set_source(g["loaded_lnid"], g["forces"])
    exp = g["expected_every20"]
    scale = np.abs(exp[:, :, 1:]).max()
    ck = {int(st): k for k, st in enumerate(g["ckpt_steps"])}
    for i in range(exp.shape[1]):
        step = 20 * i
        if step:
            s.run(20)
        if step in ck:
            tm1, tm2 = s.download()
            assert H.rel_linf(tm1, g["ckpt_tm1"][ck[step]]) < TOL
            assert H.rel_linf(tm2, g["ckpt_tm2"][ck[step]]) < TOL
        u, _ = s.gather(ids)
        st = np.einsum("sn,snd->sd", phi, u.reshape(len(phi), 8, 3))
        assert np.abs(st - exp[:, i, 1:]).max() <= 6e-7 * scale, step
    s.close()


def test_phase_force_and_update_match_reference_loops():
    """hq_phase_force == compute_addforce_effective + damping_addforce;
    hq_phase_update == solver_compute_displacement, on random fields."""
    lnid, node_ijk, elem_ijk, et, nt = _box(12, 10, 6)
    N = len(node_ijk)
    rng = np.random.default_rng(12345)
    u1 = rng.uniform(-1, 1, (N, 3)) * 1e-3
    u2 = u1 + rng.uniform(-1, 1, (N, 3)) * 1e-5
    K1, K2 = ho.compute_K()
    f = np.zeros((N, 3))
    ho.lib().ho_addforce_effective(ho.ctypes.c_int64(len(lnid)), ho._p(lnid), ho._p(et), ho._p(u1), ho._p(f), 1)
    ho.lib().ho_damping_addforce(ho.ctypes.c_int64(len(lnid)), ho._p(lnid), ho._p(et), ho._p(u1), ho._p(u2),
                                 ho._p(K1), ho._p(K2), ho._p(f), 1)
    s = ha.Solver(lnid, et, nt, 1e-3, tm1=u1, tm2=u2, variant=ha.HQ_VARIANT_SCATTER)
    s.phase_force()
    fg = s.download_force()
    assert H.rel_linf(fg, f) < 1e-12
    # update
    new = u2.copy()
    fo = f.copy()
    ho.lib().ho_compute_displacement(ho.ctypes.c_int64(N), ho._p(nt), ho._p(u1), ho._p(new), ho._p(fo), None)
    s.phase_update()
    tm1, tm2 = s.download()           # post-swap view: tm1 = newest
    assert H.rel_linf(tm1, new) < 1e-12
    assert np.array_equal(tm2, u1)
    assert np.all(s.download_force() == 0.0)
    s.close()


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("shape", [(32, 32, 16), (20, 12, 9), (1, 1, 1), (3, 1, 2)])
def test_random_field_steps_against_oracle(variant, shape):
    """Seeded random start (SURVEY s8d stress input), ragged and tiny boxes, 12 steps."""
    nx, ny, nz = shape
    lnid, node_ijk, elem_ijk, et, nt = _box(nx, ny, nz, h=10.0, dt=2e-4)
    N = len(node_ijk)
    rng = np.random.default_rng(12345)
    u1 = rng.uniform(-1, 1, (N, 3)) * 1e-3
    u2 = u1 + rng.uniform(-1, 1, (N, 3)) * 1e-6
    o1, o2 = u2.copy(), u1.copy()      # oracle arrays are pre-swap
    ho.solver_run(lnid, et, nt, o1, o2, 0, 12, 2e-4)
    s = ha.Solver(lnid, et, nt, 2e-4, tm1=u1, tm2=u2, node_xyz=_ticks(node_ijk, 1 << 20), variant=variant)
    s.run(12)
    tm1, tm2 = s.download()
    assert H.rel_linf(tm1, o2) < TOL
    assert H.rel_linf(tm2, o1) < TOL
    s.close()


def test_patch_without_coordinates_and_heterogeneous_material():
    """Patch cuts fall back to fixed runs when no coordinates are given; material
    varies per element (layered Vs, a soft pocket that trips the Vp/Vs cap)."""
    nx, ny, nz = 24, 16, 12
    elem_ijk, lnid, node_ijk = ho.uniform_mesh(nx, ny, nz)
    edata = np.empty((len(lnid), 4), np.float32)
    z = elem_ijk[:, 2]
    edata[:, 0] = 25.0
    edata[:, 2] = 800.0 + 150.0 * z
    edata[:, 1] = 1.9 * edata[:, 2]
    edata[:, 3] = 2000.0 + 30.0 * z
    soft = (elem_ijk[:, 0] < 4) & (z < 3)
    edata[soft, 2] = 300.0
    edata[soft, 1] = 1500.0            # Vp/Vs = 5 > threshold 3
    face = ho.face_bits(elem_ijk, nx, ny, nz)
    et, nt = ho.solver_init(lnid, edata, face, len(node_ijk), 1e-3, 2.0)
    N = len(node_ijk)
    rng = np.random.default_rng(7)
    u1 = rng.uniform(-1, 1, (N, 3)) * 1e-3
    u2 = u1.copy()
    o1, o2 = u2.copy(), u1.copy()
    ho.solver_run(lnid, et, nt, o1, o2, 0, 10, 1e-3)
    for variant, xyz in ((ha.HQ_VARIANT_PATCH, None), (ha.HQ_VARIANT_SCATTER, None)):
        s = ha.Solver(lnid, et, nt, 1e-3, tm1=u1, tm2=u2, node_xyz=xyz, variant=variant)
        s.run(10)
        tm1, tm2 = s.download()
        assert H.rel_linf(tm1, o2) < TOL
        s.close()


def test_hanging_node_adjust_and_roundtrip():
    """compute_adjust (DISTRIBUTION / ASSIGNMENT) with a synthetic dangling table,
    hq_upload/hq_download/hq_gather round trips."""
    lnid, node_ijk, elem_ijk, et, nt = _box(6, 6, 4, h=10.0, dt=2e-4)
    N = len(node_ijk)
    rng = np.random.default_rng(3)
    dn_ids = np.array([5, 17, 33, 60], np.int32)
    dn_ptr = np.array([0, 2, 6, 8, 12], np.int32)
    anchors = rng.choice(np.setdiff1d(np.arange(N), dn_ids), 12, replace=False).astype(np.int32)
    u1 = rng.uniform(-1, 1, (N, 3)) * 1e-3
    u2 = u1 + rng.uniform(-1, 1, (N, 3)) * 1e-6
    s = ha.Solver(lnid, et, nt, 2e-4, tm1=u1, tm2=u2, dangling=(dn_ids, dn_ptr, anchors),
                  variant=ha.HQ_VARIANT_SCATTER)
    s.run(5)
    tm1, tm2 = s.download()
    # oracle: same loop with compute_adjust around the update (psolve.c:4299, 4313)
    L = ho.lib()
    a, b = u1.copy(), u2.copy()        # post-swap view
    f = np.zeros((N, 3))
    K1, K2 = ho.compute_K()
    for _ in range(5):
        L.ho_addforce_effective(ho.ctypes.c_int64(len(lnid)), ho._p(lnid), ho._p(et), ho._p(a), ho._p(f), 1)
        L.ho_damping_addforce(ho.ctypes.c_int64(len(lnid)), ho._p(lnid), ho._p(et), ho._p(a), ho._p(b),
                              ho._p(K1), ho._p(K2), ho._p(f), 1)
        L.ho_compute_adjust(ho._p(f), 3, 0, len(dn_ids), ho._p(dn_ids), ho._p(dn_ptr), ho._p(anchors))
        L.ho_compute_displacement(ho.ctypes.c_int64(N), ho._p(nt), ho._p(a), ho._p(b), ho._p(f), None)
        L.ho_compute_adjust(ho._p(b), 3, 1, len(dn_ids), ho._p(dn_ids), ho._p(dn_ptr), ho._p(anchors))
        a, b = b, a
    assert H.rel_linf(tm1, a) < TOL
    assert H.rel_linf(tm2, b) < TOL
    g1, g2 = s.gather(np.array([0, 7, N - 1]))
    assert np.array_equal(g1, tm1[[0, 7, N - 1]]) and np.array_equal(g2, tm2[[0, 7, N - 1]])
    s.upload(u1, u2, 0)
    x1, x2 = s.download()
    assert np.array_equal(x1, u1) and np.array_equal(x2, u2)
    s.close()


@pytest.mark.parametrize("overlap", [0, 1])
@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("nranks", [2, 8])
def test_partitioned_box_matches_single_partition(variant, nranks, overlap, monkeypatch):
    """The same box cut into octor-style block partitions (C host side), all
    partitions stepped in one process on one GPU with the in-process halo
    transport: every harbored copy must agree with the oracle's single-rank run.
    overlap = 1 (HQ_OVERLAP=1): the exchange chain on its own stream beside the interior patches,
    as between GPUs (the in-process default keeps one stream per partition)."""
    from hercules_amd import capi, host
    monkeypatch.setenv("HQ_OVERLAP", str(overlap))
    nx, ny, nz, h, dt, freq = 32, 16, 16, 20.0, 4e-4, 20.0
    nsteps = 25
    boxes = [host.Box(nx, ny, nz, h, dt, freq, rank=r, nranks=nranks) for r in range(nranks)]
    # oracle, one rank
    elem_ijk, lnid, node_ijk = ho.uniform_mesh(nx, ny, nz)
    edata = np.empty((len(lnid), 4), np.float32)
    edata[:] = (h, 6000.0, 3464.0, 2700.0)
    et, nt = ho.solver_init(lnid, edata, ho.face_bits(elem_ijk, nx, ny, nz), len(node_ijk), dt, freq)
    N = len(node_ijk)
    rng = np.random.default_rng(99)
    u1 = rng.uniform(-1, 1, (N, 3)) * 1e-3
    u2 = u1 + rng.uniform(-1, 1, (N, 3)) * 1e-6
    gidx = {tuple(v): i for i, v in enumerate(node_ijk.tolist())}
    # a point source in the middle: only the rank holding the element loads it
    src = [b.point_source(nx * h / 2 + 3.0, ny * h / 2 - 2.0, nz * h / 3, 30.0, 70.0, 10.0) for b in boxes]
    owner_rank = [r for r in range(nranks) if len(src[r][0])]
    assert len(owner_rank) == 1
    rp = boxes[owner_rank[0]].run_params(loaded=src[owner_rank[0]][0], pattern=src[owner_rank[0]][1],
                                         moment=1e13, rise_time=10 * dt)
    F = boxes[owner_rank[0]].source_table(rp, 0, nsteps)
    loaded_global = [gidx[tuple(boxes[owner_rank[0]].node_ijk[i])] for i in src[owner_rank[0]][0]]
    o1, o2 = u2.copy(), u1.copy()
    ho.solver_run(lnid, et, nt, o1, o2, 0, nsteps, dt, loaded_lnid=np.array(loaded_global, np.int32), forces=F)
    solvers = []
    maps = []
    for r, b in enumerate(boxes):
        m = np.array([gidx[tuple(v)] for v in b.node_ijk.tolist()])
        maps.append(m)
        s = b.create_solver(variant=variant, tm1=u1[m], tm2=u2[m])
        assert s.info()["variant"] == variant
        if r == owner_rank[0]:
            s.set_source(src[r][0], F)
        solvers.append(s)
    capi.group_link(solvers)
    capi.group_run(solvers, nsteps)
    for r, s in enumerate(solvers):
        tm1, tm2 = s.download()
        assert H.rel_linf(tm1, o2[maps[r]]) < TOL, (r, "tm1")
        assert H.rel_linf(tm2, o1[maps[r]]) < TOL, (r, "tm2")
    for s in solvers:
        s.close()
    for b in boxes:
        b.close()


def test_restart_from_reference_checkpoint_and_force_file(tmp_path):
    """A checkpoint file and a force file laid out as the reference writes them
    (io_checkpoint.c:63-118, quakesource.c:2453-2466), built from the reference's own
    data: restart at step 400, march to 800 through the C host's solver_run, land on
    the reference's step-800 checkpoint; then write our checkpoint and re-read it."""
    from hercules_amd import host
    g = H.load("c1_short")
    p = H.c1_problem()
    N = p["N"]
    ck = tmp_path / "checkpoint.in"
    ck.write_bytes(np.array([1, 400, N], "<i4").tobytes() + g["ckpt_tm2"][0].astype("<f8").tobytes()
                   + g["ckpt_tm1"][0].astype("<f8").tobytes())
    ff = tmp_path / "force_process.0"
    host.forcefile_write(str(ff), g["loaded_lnid"], g["forces"])
    box = host.Box(H.C1_NX, H.C1_NY, H.C1_NZ, H.C1_H, 1e-3, 5.0)
    s = box.create_solver()
    assert host.checkpoint_read(s, str(ck)) == 400
    assert s.info()["step"] == 400
    ids, nsteps = host.forcefile_info(str(ff))
    rp = box.run_params(loaded=ids, force_file=str(ff), source_window=64)
    box.solver_run(s, rp, 400, 400)
    tm1, tm2 = s.download()
    assert H.rel_linf(tm1, g["ckpt_tm1"][1]) < TOL and H.rel_linf(tm2, g["ckpt_tm2"][1]) < TOL
    out = tmp_path / "checkpoint.out0"
    host.checkpoint_write(s, str(out), 800)
    b = out.read_bytes()
    assert list(np.frombuffer(b[:12], "<i4")) == [1, 800, N] and len(b) == 12 + 2 * N * 24
    assert np.array_equal(np.frombuffer(b[12:12 + N * 24], "<f8").reshape(N, 3), tm2)
    assert np.array_equal(np.frombuffer(b[12 + N * 24:], "<f8").reshape(N, 3), tm1)
    s2 = box.create_solver(variant=ha.HQ_VARIANT_SCATTER)
    assert host.checkpoint_read(s2, str(out)) == 800
    x1, x2 = s2.download()
    assert np.array_equal(x1, tm1) and np.array_equal(x2, tm2)
    s.close(); s2.close(); box.close()


@pytest.mark.parametrize("mesh", ["c5_two_level", "c5_three_level", "c5_layered", "c5_basin", "c5_gradient"])
@pytest.mark.parametrize("variant", VARIANTS + [ha.HQ_VARIANT_AUTO])
def test_two_level_mesh_with_hanging_nodes_against_reference(variant, mesh):
    """compute_adjust on the reference's own two-level mesh (800 hanging nodes): scatter
    kernels + adjust kernels, and the patch kernel (hanging-node forces accumulated by the
    patches that own their anchors, assignment kernel after), against the checkpoints the
    real reference wrote (also its three-level, three-material mesh, and c5_basin: the LATERALLY
    refined mesh -- level interfaces with x-, y- and z-normal faces, staircase corners, hanging
    nodes of every orientation, on domain faces too)."""
    p = H.c5_problem(mesh)
    g = p["golden"]
    s = ha.Solver(p["lnid"], p["etable"], p["ntable"], p["dt"], dangling=p["dangling"], variant=variant,
                  node_xyz=(p["node_q"].astype(np.int64) * p["emin"]).astype(np.int32))
    assert s.info()["variant"] == (ha.HQ_VARIANT_PATCH if variant == ha.HQ_VARIANT_AUTO else variant)
    s.set_source(g["loaded_lnid"], g["forces"])
    done = 0
    for k, step in enumerate(g["ckpt_steps"]):
        s.run(int(step) - done)
        done = int(step)
        tm1, tm2 = s.download()
        assert H.rel_linf(tm1, g["ckpt_tm1"][k]) < TOL
        assert H.rel_linf(tm2, g["ckpt_tm2"][k]) < TOL
    s.close()


def test_ragged_tile_columns_on_the_references_lateral_mesh_against_its_checkpoints():
    """c5_basin -- the mesh the REAL reference refined laterally, hanging nodes of every orientation -- with the planner's
    thresholds lowered (hq_options: 12 nodes per plane, 48 per column, 2 planes) until this 5 429-element mesh carries ragged
    tile columns (HQ_BK_RAGGED) beside its x-, y- and z-normal level interfaces: against the reference's own checkpoints."""
    p = H.c5_problem("c5_basin")
    g = p["golden"]
    s = ha.Solver(p["lnid"], p["etable"], p["ntable"], p["dt"], dangling=p["dangling"], variant=ha.HQ_VARIANT_PATCH,
                  node_xyz=(p["node_q"].astype(np.int64) * p["emin"]).astype(np.int32),
                  options={"brick_ragged_minfill": 12, "brick_minnodes": 48, "brick_minz": 2})
    import os
    assert s.info()["brick_units_ragged"] >= (0 if os.environ.get("HQ_NO_BRICKS") else 2)      # (this file runs twice: see its fixture)
    s.set_source(g["loaded_lnid"], g["forces"])
    done = 0
    for k, step in enumerate(g["ckpt_steps"]):
        s.run(int(step) - done)
        done = int(step)
        tm1, tm2 = s.download()
        assert H.rel_linf(tm1, g["ckpt_tm1"][k]) < TOL
        assert H.rel_linf(tm2, g["ckpt_tm2"][k]) < TOL
    s.close()


@pytest.mark.parametrize("pack", [1, 0])
def test_ragged_per_element_units_on_the_references_gradient_mesh_against_its_checkpoints(pack):
    """c5_gradient -- the reference's laterally refined basin with a material of its own in every database octant -- with
    the planner's thresholds lowered until the small mesh carries RAGGED units of the per-element kernel
    (hq_k_brick_het<PACKED, RAGGED> with hq_desc.edata, <false, RAGGED> without): against the reference's own checkpoints."""
    import os
    p = H.c5_problem("c5_gradient")
    g = p["golden"]
    kw = dict(edata=p["edata"], material=H.c5_material(p)) if pack else {}      # (edata_t as solver_init left it)
    s = ha.Solver(p["lnid"], p["etable"], p["ntable"], p["dt"], dangling=p["dangling"], variant=ha.HQ_VARIANT_PATCH,
                  node_xyz=(p["node_q"].astype(np.int64) * p["emin"]).astype(np.int32),
                  options={"brick_ragged_minfill": 12, "brick_minnodes": 48, "brick_minz": 2}, **kw)
    info = s.info()
    if not os.environ.get("HQ_NO_BRICKS"):                   # (this file runs twice: see its fixture)
        assert info["brick_units_ragged_het"] >= 2 and info["brick_units_het"] >= info["brick_units_ragged_het"]
        assert (info["brick_units_packed"] > 0) == bool(pack)
    s.set_source(g["loaded_lnid"], g["forces"])
    done = 0
    for k, step in enumerate(g["ckpt_steps"]):
        s.run(int(step) - done)
        done = int(step)
        tm1, tm2 = s.download()
        assert H.rel_linf(tm1, g["ckpt_tm1"][k]) < TOL
        assert H.rel_linf(tm2, g["ckpt_tm2"][k]) < TOL
    s.close()


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("shape", [(64, 64, 8, 12), (128, 96, 16, 20)])
def test_larger_two_level_meshes_against_oracle(variant, shape):
    """Hanging nodes at non-trivial size (up to ~0.3M elements, 12k dangling nodes): patch
    cuts across the refinement interface, extra accumulators, in-LDS distribution."""
    nx, ny, nzf, nzc = shape
    p = H.two_level_mesh(nx, ny, nzf, nzc, h_fine=10.0, dt=2e-4, freq=20.0)
    N = p["N"]
    rng = np.random.default_rng(2024)
    u1 = rng.uniform(-1, 1, (N, 3)) * 1e-3
    u2 = u1 + rng.uniform(-1, 1, (N, 3)) * 1e-6
    ho.compute_adjust(u1, 1, p["dangling"])            # a consistent state: hanging = mean of anchors
    ho.compute_adjust(u2, 1, p["dangling"])
    nsteps = 6
    o1, o2 = u2.copy(), u1.copy()
    ho.solver_run(p["lnid"], p["etable"], p["ntable"], o1, o2, 0, nsteps, p["dt"], dangling=p["dangling"])
    s = ha.Solver(p["lnid"], p["etable"], p["ntable"], p["dt"], tm1=u1, tm2=u2, dangling=p["dangling"],
                  node_xyz=p["node_q"], variant=variant)
    s.run(nsteps)
    tm1, tm2 = s.download()
    assert H.rel_linf(tm1, o2) < TOL and H.rel_linf(tm2, o1) < TOL
    s.close()


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("name", ["c5_two_level_np8", "c5_basin_np8", "c5_basin_np5", "c5_gradient_np8"])
def test_octree_mesh_on_eight_partitions_against_reference(variant, name):
    """The reference's 8-rank run of its two-level mesh: hanging nodes shared between
    ranks (dn_sched), anchors harbored indirectly, the four exchanges of a step and both
    compute_adjust passes -- eight contexts stepped in one process on one GPU against the
    reference's per-rank checkpoint stripes.  Scatter kernels, and the patch kernel with the
    interface table carrying shared hanging nodes and their anchors.  c5_basin_np8 / _np5: the
    laterally refined basin as the reference ran it on 8 and on 5 ranks (hanging nodes with anchors
    on other ranks across x- and y-normal level interfaces)."""
    from hercules_amd import capi
    pr = H.c5_np8_problem(name)
    g, parts = pr["golden"], pr["parts"]
    mesh = pr["mesh"]
    solvers = []
    for p in parts:
        r = p["rank"]
        s = ha.Solver(p["lnid"], pr["ets"][r], pr["nts"][r], pr["dt"], dangling=p["dangling"],
                      an_sched=p["an_sched"], dn_sched=p["dn_sched"], rank=r, nranks=pr["nranks"], variant=variant,
                      node_xyz=mesh["node_q"][p["nodes"]])
        assert s.info()["variant"] == variant
        if len(pr["loaded"][r]):
            s.set_source(pr["loaded"][r], pr["forces"][r])
        solvers.append(s)
    capi.group_link(solvers)
    done = 0
    for step in g["ckpt_steps"]:
        capi.group_run(solvers, int(step) - done)
        done = int(step)
        for p, s in zip(parts, solvers):
            ref2, ref1 = H.np8_stripe(g, step, p["rank"], len(p["nodes"]))
            tm1, tm2 = s.download()
            scale = max(np.abs(ref1).max(), 1.0)
            assert np.abs(tm1 - ref1).max() <= TOL * scale and np.abs(tm2 - ref2).max() <= TOL * scale
    for s in solvers:
        s.close()


def test_output_planes_written_by_the_c_solver_run(tmp_path):
    """hqh_solver_run with two output planes, driven by the reference's own force file: the files
    equal planedisplacements.<i> as the REAL reference wrote them (tests/golden/c1_planes.npz)."""
    from hercules_amd import host
    g = H.load("c1_planes")
    box = host.Box(H.C1_NX, H.C1_NY, H.C1_NZ, H.C1_H, 1e-3, 5.0)
    lonc, latc = g["surface_corners_lon_lat"][:, 0], g["surface_corners_lon_lat"][:, 1]
    planes = []
    for spec in g["plane_specs"]:
        lat, lon, depth, ds, ns, dd, nd, strike, dip = spec
        x, y = host.domain_coords(lon, lat, lonc, latc, g["domain_xyz"][0], g["domain_xyz"][1])
        pts = host.plane_points((x, y, depth), ds, int(ns), dd, int(nd), strike, dip)
        ids, phi, mine = box.stations(pts)
        assert mine.all()
        planes.append((ids, phi))
    ff = tmp_path / "force_process.0"
    host.forcefile_write(str(ff), g["loaded_lnid"], g["forces"])
    nsteps = int(round(float(g["end_time"]) / float(g["dt"])))
    s = box.create_solver()
    rp = box.run_params(loaded=g["loaded_lnid"], force_file=str(ff), source_window=64, planes=planes,
                        plane_rate=int(g["plane_rate"]), plane_dir=str(tmp_path))
    box.solver_run(s, rp, 0, 150)            # in two calls: the second one appends
    box.solver_run(s, rp, 150, nsteps - 150)
    s.close()
    for i, (ids, _) in enumerate(planes):
        got = np.fromfile(tmp_path / ("planedisplacements.%d" % i), "<f8").reshape(-1, len(ids), 3)
        ref = g["plane%d" % i]
        assert got.shape == ref.shape
        assert H.rel_linf(got, ref) < TOL


@pytest.mark.parametrize("overlap", [0, 1])
@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("shape,nranks", [((16, 8, 6, 3), 5), ((32, 32, 4, 6), 8)])
def test_partitioned_two_level_box_of_the_c_host(variant, shape, nranks, overlap, monkeypatch):
    """The two-level box with hanging nodes cut into octor's partitions by the C host
    (hqh_octbox_create, nranks > 1), all partitions stepped in one process with the in-process
    transport (all four exchanges of a step): every harbored copy equals the oracle's run of the
    whole box."""
    from hercules_amd import capi, host
    monkeypatch.setenv("HQ_OVERLAP", str(overlap))      # 1: exchange chain on its own stream (hanging nodes included)
    nx, ny, nzf, nzc = shape
    nsteps = 20
    ref = H.two_level_mesh(nx, ny, nzf, nzc)
    N = ref["N"]
    rng = np.random.default_rng(7)
    u1 = rng.uniform(-1, 1, (N, 3)) * 1e-3
    u2 = u1 + rng.uniform(-1, 1, (N, 3)) * 1e-6
    # hanging nodes carry the mean of their anchors (compute_adjust ASSIGNMENT)
    ho.compute_adjust(u1, 1, ref["dangling"])
    ho.compute_adjust(u2, 1, ref["dangling"])
    o1, o2 = u2.copy(), u1.copy()
    ho.solver_run(ref["lnid"], ref["etable"], ref["ntable"], o1, o2, 0, nsteps, ref["dt"], dangling=ref["dangling"])
    boxes = [host.OctBox(nx, ny, nzf, nzc, 31.25, ref["dt"], 5.0, rank=r, nranks=nranks) for r in range(nranks)]
    assert sum(b.E for b in boxes) == ref["E"]
    assert sum(int((b.owner == r).sum()) for r, b in enumerate(boxes)) == N
    solvers = [b.create_solver(variant=variant, tm1=u1[b.gid], tm2=u2[b.gid]) for b in boxes]
    capi.group_link(solvers)
    capi.group_run(solvers, nsteps)
    for r, (b, s) in enumerate(zip(boxes, solvers)):
        tm1, tm2 = s.download()
        assert H.rel_linf(tm1, o2[b.gid]) < TOL, (r, "tm1")
        assert H.rel_linf(tm2, o1[b.gid]) < TOL, (r, "tm2")
    for s in solvers:
        s.close()
    for b in boxes:
        b.close()


@pytest.mark.parametrize("nranks", [1, 5])
def test_three_level_box_of_the_c_host(nranks):
    """The three-level layered box built (and cut into octor's partitions) by the C host,
    stepped on the GPU, against the oracle on the mesh the REAL reference generated for the same
    model (tests/golden/c5_three_level)."""
    from hercules_amd import capi, host
    real = H.c5_problem("c5_three_level")
    g = real["golden"]
    mats = {float(r[0]): (float(r[1]), float(r[0]), float(r[2])) for r in np.unique(g["mat_vs_vp_rho"], axis=0)}
    levels = [(2,) + mats[150.0], (1,) + mats[2000.0], (1,) + mats[3464.0]]
    N, nsteps = real["N"], 40
    rng = np.random.default_rng(11)
    u1 = rng.uniform(-1, 1, (N, 3)) * 1e-3
    u2 = u1 + rng.uniform(-1, 1, (N, 3)) * 1e-6
    ho.compute_adjust(u1, 1, real["dangling"])
    ho.compute_adjust(u2, 1, real["dangling"])
    o1, o2 = u2.copy(), u1.copy()
    ho.solver_run(real["lnid"], real["etable"], real["ntable"], o1, o2, 0, nsteps, real["dt"], dangling=real["dangling"])
    boxes = [host.OctBox(16, 16, 0, 0, 62.5, real["dt"], float(g["freq"]), levels=levels, rank=r, nranks=nranks)
             for r in range(nranks)]
    gids = [b.gid if nranks > 1 else np.arange(N) for b in boxes]
    solvers = [b.create_solver(tm1=u1[i], tm2=u2[i]) for b, i in zip(boxes, gids)]
    if nranks > 1:
        capi.group_link(solvers)
        capi.group_run(solvers, nsteps)
    else:
        solvers[0].run(nsteps)
    for r, (s, i) in enumerate(zip(solvers, gids)):
        tm1, tm2 = s.download()
        assert H.rel_linf(tm1, o2[i]) < TOL, (r, "tm1")
        assert H.rel_linf(tm2, o1[i]) < TOL, (r, "tm2")
    for s in solvers:
        s.close()
    for b in boxes:
        b.close()


def test_from_the_references_mesh_database_to_its_checkpoints(tmp_path):
    """End to end on the reference's own files: mesh.e (the mesh database its mesher wrote) read by
    the C host, tables built from the leaves, its force file streamed by hqh_solver_run -- the
    field equals the checkpoints the REAL reference wrote for that run (tests/golden/c5_layered:
    three octree levels after 2:1 balancing, 1008 hanging nodes)."""
    import bz2
    from hercules_amd import host
    g = H.load("c5_layered")
    path = tmp_path / "mesh.e"
    path.write_bytes(bz2.decompress(g["mesh_e_bz2"].tobytes()))
    ticks, level, vals = host.etree_read(str(path))
    _, edata = host.mesh_payload(vals)
    edge = np.uint32(1) << (30 - level).astype(np.uint32)
    ob = host.OctBox.from_leaves(ticks, edge, edata, H.C1_FAR_TICKS, float(g["dt"]), float(g["freq"]))
    ff = tmp_path / "force_process.0"
    host.forcefile_write(str(ff), g["loaded_lnid"], g["forces"])
    s = ob.create_solver()
    rp = ob.run_params(loaded=g["loaded_lnid"], force_file=str(ff), source_window=32)
    done = 0
    for k, step in enumerate(g["ckpt_steps"]):
        ob.solver_run(s, rp, done, int(step) - done)        # hqh_octbox_solver_run: the force file in windows
        done = int(step)
        tm1, tm2 = s.download()
        assert H.rel_linf(tm1, g["ckpt_tm1"][k]) < TOL and H.rel_linf(tm2, g["ckpt_tm2"][k]) < TOL
    s.close()
    ob.close()


@pytest.mark.parametrize("name", ["c5_basin", "c5_gradient"])
def test_from_the_cvm_database_to_the_references_checkpoints(name, tmp_path):
    """End to end from "the same input etree" WITHOUT the reference's mesher (SURVEY s8 f4's parenthetical, round 6): the
    CVM database of the golden run -- written here by oracle/make_cvm on the reference's own etree + cvm libraries, as
    tests/golden/make_golden.py did -- read by hqh_cvm_open, meshed by hqh_octree_generate (Vs rule on setrec's 27-sample
    record, 2:1 balance), tables by hqh_mesh_from_leaves, the reference's force file streamed by hqh_solver_run: the field
    lands on the checkpoints the REAL psolve wrote from that database (laterally refined basin; the same with a
    material of its own in every database octant)."""
    import os
    import subprocess
    from hercules_amd import host
    from tests.test_host_partition import MAKE_CVM, _cvm_args
    if not os.path.exists(MAKE_CVM):
        pytest.skip("oracle/_ref/make_cvm is not built")
    db = str(tmp_path / "model.e")
    subprocess.run([MAKE_CVM, db] + _cvm_args(name), check=True)
    cvm = host.Cvm(db)
    vp, vs, rho, cell = cvm.grid()
    region = cvm.region
    cvm.close()
    g = H.load(name)
    spec = H.CVM_MODELS[name]
    # (the mesh's x is the database's north: setrec queries east = y, north = x)
    ticks, edge, edata, far, ticksize = host.octree_generate(vp, vs, rho, cell, (region[1], region[0], region[2]),
                                                             spec["freq"] * 8, spec["vscut"])
    assert len(ticks) == int(g["total_elements"])
    ob = host.OctBox.from_leaves(ticks, edge, edata, far, float(g["dt"]), float(g["freq"]))
    assert ob.N == int(g["total_nodes"]) and ob.ldnnum == int(g["total_dangling"])
    ff = tmp_path / "force_process.0"
    host.forcefile_write(str(ff), g["loaded_lnid"], g["forces"])
    s = ob.create_solver()
    rp = ob.run_params(loaded=g["loaded_lnid"], force_file=str(ff), source_window=32)
    done = 0
    for k, step in enumerate(g["ckpt_steps"]):
        ob.solver_run(s, rp, done, int(step) - done)
        done = int(step)
        tm1, tm2 = s.download()
        assert H.rel_linf(tm1, g["ckpt_tm1"][k]) < TOL and H.rel_linf(tm2, g["ckpt_tm2"][k]) < TOL
    s.close()
    ob.close()


def test_partitions_cut_from_a_mesh_database(tmp_path):
    """The reference's layered-model mesh read from its mesh.e, cut into 5 octor partitions by the
    C host (general octree: Z-order point location), stepped in one process with the in-process
    transport, against the oracle's run of the whole mesh."""
    import bz2
    from hercules_amd import capi, host
    g = H.load("c5_layered")
    real = H.c5_problem("c5_layered")
    path = tmp_path / "mesh.e"
    path.write_bytes(bz2.decompress(g["mesh_e_bz2"].tobytes()))
    ticks, level, vals = host.etree_read(str(path))
    _, edata = host.mesh_payload(vals)
    edge = np.uint32(1) << (30 - level).astype(np.uint32)
    N, nsteps, nranks = real["N"], 30, 5
    rng = np.random.default_rng(5)
    u1 = rng.uniform(-1, 1, (N, 3)) * 1e-3
    u2 = u1 + rng.uniform(-1, 1, (N, 3)) * 1e-6
    ho.compute_adjust(u1, 1, real["dangling"])
    ho.compute_adjust(u2, 1, real["dangling"])
    o1, o2 = u2.copy(), u1.copy()
    ho.solver_run(real["lnid"], real["etable"], real["ntable"], o1, o2, 0, nsteps, real["dt"], dangling=real["dangling"])
    boxes = [host.OctBox.from_leaves(ticks, edge, edata, H.C1_FAR_TICKS, real["dt"], float(g["freq"]), rank=r,
                                     nranks=nranks) for r in range(nranks)]
    solvers = [b.create_solver(tm1=u1[b.gid], tm2=u2[b.gid]) for b in boxes]
    capi.group_link(solvers)
    capi.group_run(solvers, nsteps)
    for r, (b, s) in enumerate(zip(boxes, solvers)):
        tm1, tm2 = s.download()
        assert H.rel_linf(tm1, o2[b.gid]) < TOL and H.rel_linf(tm2, o1[b.gid]) < TOL, r
    for s in solvers:
        s.close()
    for b in boxes:
        b.close()


def test_checkpoints_at_the_references_cadence(tmp_path):
    """hqh_solver_run with checkpointing_rate = 400 on examples/simple: checkpoint.out0 (step 400)
    and checkpoint.out1 (step 800) as the REAL reference wrote them (tests/golden/c1_short)."""
    from hercules_amd import host
    g = H.load("c1_short")
    N = H.c1_problem()["N"]
    ff = tmp_path / "force_process.0"
    host.forcefile_write(str(ff), g["loaded_lnid"], g["forces"])
    box = host.Box(H.C1_NX, H.C1_NY, H.C1_NZ, H.C1_H, 1e-3, 5.0)
    s = box.create_solver()
    rp = box.run_params(loaded=g["loaded_lnid"], force_file=str(ff), source_window=128, checkpoint_rate=400,
                        checkpoint_dir=str(tmp_path))
    box.solver_run(s, rp, 0, 1000)
    s.close()
    assert list(g["ckpt_steps"]) == [400, 800]
    for k, name in enumerate(("checkpoint.out0", "checkpoint.out1")):
        b = (tmp_path / name).read_bytes()
        assert list(np.frombuffer(b[:12], "<i4")) == [1, int(g["ckpt_steps"][k]), N] and len(b) == 12 + 2 * N * 24
        tm2 = np.frombuffer(b[12:12 + N * 24], "<f8").reshape(N, 3)
        tm1 = np.frombuffer(b[12 + N * 24:], "<f8").reshape(N, 3)
        assert H.rel_linf(tm1, g["ckpt_tm1"][k]) < TOL and H.rel_linf(tm2, g["ckpt_tm2"][k]) < TOL


def test_rccl_binding_on_a_communicator_of_one():
    """The multi-GPU transport on a one-GPU box: librccl is loaded, a communicator of one rank is
    created through hq_comm_unique_id / hq_comm_init, and hq_comm_selftest moves doubles from the
    rank to itself with the grouped ncclRecv/ncclSend the halo exchange issues (same stream, same
    argument order).  The exchange logic itself is covered by the partitioned-box tests over the
    in-process transport and by the gloo tests on the host side."""
    lnid, node_ijk, elem_ijk, et, nt = _box(4, 4, 4)
    s = ha.Solver(lnid, et, nt, 1e-3, node_xyz=_ticks(node_ijk, 1 << 20))
    s.comm_init(ha.capi.comm_unique_id())
    s.comm_selftest(3 * 4096)
    with pytest.raises(ha.capi.HqError):
        s.comm_init(ha.capi.comm_unique_id())      # a second communicator is refused
    s.run(3)                                        # no neighbours: the run is untouched by the communicator
    s.close()


@pytest.mark.parametrize("variant", [ha.HQ_VARIANT_PATCH])
def test_mid_size_box_against_the_references_checkpoint(variant):
    """the reference's mesher refines examples/simple to 128 x 128 x 64 =
    1 048 576 elements at f = 40 Hz, dt = 0.5 ms; tests/golden/c2_mid holds its force file and its
    step-200 checkpoint at 4 352 nodes (seeded sample + the largest amplitudes), each with the
    node's tick coordinates from the reference's mesh.e.  The C host side builds the same box
    (hqh_box_create restates solver_init), nodes are matched BY COORDINATE, and octor's numbering
    is then asserted to be the one the box has."""
    from hercules_amd import host as hhost
    g = H.load("c2_mid")
    nx, ny, nz = 128, 128, 64
    box = hhost.Box(nx, ny, nz, 1000.0 / nx, float(g["dt"]), float(g["freq"]))
    assert box.info["lenum"] == int(g["elements"]) and box.info["nharbored"] == int(g["nodes"])
    edge = int(g["edge_ticks"])

    def key(ijk):
        ijk = np.asarray(ijk, np.int64)
        return ijk[:, 0] + (nx + 1) * (ijk[:, 1] + (ny + 1) * ijk[:, 2])

    lut = np.full((nx + 1) * (ny + 1) * (nz + 1), -1, np.int64)
    lut[key(box.node_ijk)] = np.arange(len(box.node_ijk))
    assert (g["sample_ticks"] % edge == 0).all()
    sample = lut[key(g["sample_ticks"] // edge)]
    loaded = lut[key(g["loaded_ticks"] // edge)]
    assert (sample >= 0).all() and (loaded >= 0).all()
    assert np.array_equal(sample, g["sample_lnid"])           # octor's node order (octor.c:6166) is the box's
    assert np.array_equal(loaded, g["loaded_lnid"])
    s = box.create_solver(variant=variant)
    s.set_source(loaded.astype(np.int32), g["forces"])
    s.run(int(g["ckpt_step"]))
    tm1, tm2 = s.gather(sample.astype(np.int32))
    e1 = np.abs(tm1 - g["sample_tm1"]).max() / float(g["max_abs_tm1"])
    e2 = np.abs(tm2 - g["sample_tm2"]).max() / float(g["max_abs_tm2"])
    assert e1 < TOL and e2 < TOL, (e1, e2)
    f1, f2 = s.download()
    assert abs(np.abs(f1).max() / float(g["max_abs_tm1"]) - 1.0) < TOL
    assert np.abs(f1.sum(axis=0) - g["sum_tm1"]).max() <= 1e-9 * np.abs(g["sum_abs_tm1"]).max()
    s.close()
    box.close()


def test_station_files_with_velocities_and_accelerations(tmp_path):
    """hqh_solver_run with station_derivs = 2 on examples/simple, driven by the reference's force
    file: displacement, velocity and acceleration of the five stations at every step, against the
    station files the reference wrote with print_station_velocities / _accelerations = yes
    (tests/golden/c1_stations_va; printed with 7 digits).  The acceleration needs u(t - 2 dt):
    hq_gather3 reads it from the buffer the next step overwrites."""
    from hercules_amd import host
    g = H.load("c1_stations_va")
    ref = g["stations"]                                  # [5, steps, 10]
    steps = ref.shape[1]
    ff = tmp_path / "force_process.0"
    host.forcefile_write(str(ff), g["loaded_lnid"], g["forces"])
    box = host.Box(H.C1_NX, H.C1_NY, H.C1_NZ, H.C1_H, 1e-3, 5.0)
    ids, phi, mine = box.stations(H.C1_STATIONS)
    assert mine.all()
    s = box.create_solver()
    got = np.zeros((steps, len(ids), 9))
    text = [host.station_header(2)]

    def on_print(step, vals):
        got[step] = vals
        text.append(host.station_format(step * 1e-3, vals[0]))

    rp = box.run_params(loaded=g["loaded_lnid"], force_file=str(ff), source_window=128, station_ids=ids,
                        station_phi=phi, station_rate=1, station_fn=on_print, station_derivs=2)
    box.solver_run(s, rp, 0, 250)                       # in two calls, as a restartable driver would
    box.solver_run(s, rp, 250, steps - 250)
    for k in range(3):
        scale = np.abs(ref[:, :, 1 + 3 * k:4 + 3 * k]).max()
        err = np.abs(got[:, :, 3 * k:3 * k + 3].transpose(1, 0, 2) - ref[:, :, 1 + 3 * k:4 + 3 * k]).max()
        assert err <= 6e-7 * scale, (k, err, scale)
    # the text itself: header and the quiet first lines are the reference's bytes
    ref_lines = str(g["station0_text"]).split("\n")
    ours = "".join(text).split("\n")
    assert ours[0] == ref_lines[0]
    assert ours[1:4] == ref_lines[1:4]
    # ... and every printed number that is not rounding noise (station 0 sits on the fault plane: its
    # z components are ~1e-18 of the others) has the reference's digits, up to a rare last-digit flip
    fields = flips = 0
    for t in range(1, len(ref_lines) - 1):
        a, b = ours[t + 1].split(), ref_lines[t + 1].split()
        assert len(a) == len(b) == 10 and a[0] == b[0]
        for k in range(1, 10):
            if abs(float(b[k])) > 1e-6 * np.abs(ref[0, :, 1 + 3 * ((k - 1) // 3):4 + 3 * ((k - 1) // 3)]).max():
                fields += 1
                flips += a[k] != b[k]
    assert fields > 500 and flips <= 3, (fields, flips)
    # hq_gather3 itself: tm3 of now is tm2 of one step ago
    a1, a2 = s.gather(ids)
    s.run(1)
    b1, b2, b3 = s.gather3(ids)
    assert np.array_equal(b2, a1) and np.array_equal(b3, a2)
    sc = box.create_solver(variant=ha.HQ_VARIANT_SCATTER)
    with pytest.raises(ha.capi.HqError):
        sc.gather3(ids)
    s.close(); sc.close(); box.close()


def test_4d_wavefield_files_written_by_the_c_solver_run(tmp_path):
    """hqh_solver_run with the 4D output on (displacement and velocity of every node every 100
    steps, the reference's out_hdr_t file format), driven by the reference's force file, against
    disp.h4d / vel.h4d the reference wrote (tests/golden/c1_wavefield)."""
    from hercules_amd import host
    g = H.load("c1_wavefield")
    box = host.Box(H.C1_NX, H.C1_NY, H.C1_NZ, H.C1_H, 1e-3, 5.0)
    N, E = box.info["nharbored"], box.info["lenum"]
    ff = tmp_path / "force_process.0"
    host.forcefile_write(str(ff), g["loaded_lnid"], g["forces"])
    paths = {q: str(tmp_path / (q + ".h4d")) for q in ("disp", "vel")}
    for q, name in (("disp", "displacement"), ("vel", "velocity")):
        host.wavefield_create(paths[q], name, N, E, (1000.0, 1000.0, 500.0), 1000.0 / 2 ** 30, 1e-3, 100, 349)
    s = box.create_solver()
    rp = box.run_params(loaded=g["loaded_lnid"], force_file=str(ff), source_window=64, wavefield_rate=100,
                        wavefield_disp_file=paths["disp"], wavefield_vel_file=paths["vel"], wavefield_total_nodes=N)
    box.solver_run(s, rp, 0, 120)
    box.solver_run(s, rp, 120, 229)                     # 349 steps in all, as the reference ran
    for q in ("disp", "vel"):
        ref, ours = g[q + "_np1"].tobytes(), open(paths[q], "rb").read()
        assert len(ours) == len(ref)
        assert ours[:32] == ref[:32] and ours[48:128] == ref[48:128]
        a = np.frombuffer(ours[136:], "<f8").reshape(4, N, 3)
        b = np.frombuffer(ref[136:], "<f8").reshape(4, N, 3)
        assert not a[0].any() and not b[0].any()
        for k in range(1, 4):
            assert H.rel_linf(a[k], b[k]) < (TOL if q == "disp" else 1e-7)   # velocity: a difference of two fields / dt
    s.close(); box.close()


@pytest.mark.parametrize("variant", VARIANTS)
def test_halo_debug_mode_and_nan_probe(variant, monkeypatch):
    """HQ_DEBUG_HALO=1 is the reference's -DDEBUG exchange (psolve.c:5002-5007, 5058-5069): every halo
    record carries the global identity of its node and the receiver checks it.  A correct schedule
    passes (two-level octree box on 5 partitions: all four exchanges of a step); a schedule with two
    entries of one side swapped is caught at hq_sync.  hq_check_finite is solver_check_nan
    (psolve.c:3769-3782)."""
    from hercules_amd import capi, host
    monkeypatch.setenv("HQ_DEBUG_HALO", "1")
    nranks = 5
    boxes = [host.OctBox(16, 8, 6, 3, 31.25, 1e-3, 5.0, rank=r, nranks=nranks) for r in range(nranks)]
    ref = H.two_level_mesh(16, 8, 6, 3)
    rng = np.random.default_rng(7)
    u1 = rng.uniform(-1, 1, (ref["N"], 3)) * 1e-3
    ho.compute_adjust(u1, 1, ref["dangling"])
    solvers = [b.create_solver(variant=variant, tm1=u1[b.gid], tm2=u1[b.gid]) for b in boxes]
    capi.group_link(solvers)
    capi.group_run(solvers, 6)                       # raises on any identity mismatch
    assert all(s.check_finite() == 0 for s in solvers)
    bad = u1[boxes[0].gid].copy()
    bad[3, 1] = np.nan
    bad[5, 0] = np.inf
    solvers[0].upload(bad, bad, 0)
    assert solvers[0].check_finite() == 4            # two values, in tm1 and in tm2
    for s in solvers:
        s.close()
    # the same partitions with one rank's c-list records in another order than its owner's s-list
    sch = [b.schedules() for b in boxes]
    victim = next(r for r in range(nranks) if any(len(m) > 1 for _, m in sch[r]["an"]["c"]))
    lst = sch[victim]["an"]["c"]
    k = next(i for i, (_, m) in enumerate(lst) if len(m) > 1)
    lst[k][1][[0, 1]] = lst[k][1][[1, 0]]
    solvers = []
    for r, b in enumerate(boxes):
        solvers.append(ha.Solver(b.lnid, b.etable, b.ntable, b.dt, tm1=u1[b.gid], tm2=u1[b.gid], node_xyz=b.node_xyz,
                                 dangling=b.dangling, an_sched=sch[r]["an"], dn_sched=sch[r]["dn"], rank=r,
                                 nranks=nranks, variant=variant))
    capi.group_link(solvers)
    with pytest.raises(capi.HqError, match="HQ_DEBUG_HALO"):
        capi.group_run(solvers, 2)
    for s in solvers:
        s.close()
    for b in boxes:
        b.close()


def test_create_refuses_a_table_that_is_not_rayleigh_proportional():
    """The fused element product needs c3/c1 == c4/c2 (both are b/dt in solver_init, psolve.c:3386-3409)."""
    lnid, node_ijk, elem_ijk, et, nt = _box(4, 4, 2)
    et = et.copy()
    et[3, 3] *= 1.5
    with pytest.raises(ha.HqError, match="Rayleigh"):
        ha.Solver(lnid, et, nt, 1e-3)


@pytest.mark.parametrize("pipe", ["0", "4", "6", "4-nostencil", "6-nostencil", "6-ragged"])
def test_every_patch_kernel_on_a_partitioned_octree_box(pipe, monkeypatch, brick_mode):
    """The element-form patch kernels -- hq_k_patch_seed (default, HQ_PATCH_PIPE=6), hq_k_patch_pers (4: the form a
    mesh falls back to when three accumulator arrays do not fit LDS) and hq_k_patch_step (0: the form for
    patches of more than 1024 elements) -- with and without hq_k_patch_stencil taking the uniform lattice patches
    (HQ_PATCH_NO_STENCIL=1: the lattice patches go through the element kernels' lattice rows), on the same problem: the two-level octree box with hanging nodes
    on 5 partitions (all four exchanges of a step, interface seeds, hanging-node seeds) against the oracle's
    single-rank run, and a uniform box with lattice patches, dashpot faces and a point source."""
    from hercules_amd import capi, host
    nostencil = pipe.endswith("-nostencil")      # lattice patches through the element kernels instead of hq_k_patch_stencil
    ragged = pipe.endswith("-ragged")            # lattice-SUBSET patches (faces, dashpots) through hq_k_patch_stencil too
    monkeypatch.setenv("HQ_PATCH_RAGGED", "1" if ragged else "0")    # (the default is on)
    pipe = pipe.split("-")[0]
    monkeypatch.setenv("HQ_PATCH_PIPE", pipe)
    if nostencil:
        monkeypatch.setenv("HQ_PATCH_NO_STENCIL", "1")
    nranks, nsteps = 5, 12
    ref = H.two_level_mesh(16, 8, 6, 3)
    rng = np.random.default_rng(3)
    u1 = rng.uniform(-1, 1, (ref["N"], 3)) * 1e-3
    u2 = u1 + rng.uniform(-1, 1, (ref["N"], 3)) * 1e-6
    ho.compute_adjust(u1, 1, ref["dangling"])
    ho.compute_adjust(u2, 1, ref["dangling"])
    o1, o2 = u2.copy(), u1.copy()
    ho.solver_run(ref["lnid"], ref["etable"], ref["ntable"], o1, o2, 0, nsteps, ref["dt"], dangling=ref["dangling"])
    boxes = [host.OctBox(16, 8, 6, 3, 31.25, ref["dt"], 5.0, rank=r, nranks=nranks) for r in range(nranks)]
    solvers = [b.create_solver(variant=ha.HQ_VARIANT_PATCH, tm1=u1[b.gid], tm2=u2[b.gid]) for b in boxes]
    want = {"0": "hq_k_patch_step", "4": "hq_k_patch_pers", "6": "hq_k_patch_seed"}[pipe]
    infos = [s.info() for s in solvers]
    if nostencil:
        assert all(i["stencil_patches"] == 0 for i in infos)
    # the partitions of this small mesh are single patches of more than 512 nodes: element form (the stencil kernel's
    # interface launch is covered by the 8-partition boxes of test_gpu_fullsize.py and by the uniform box below)
    assert all(s.dominant_kernel() in (want, "hq_k_patch_stencil") for s in solvers)
    capi.group_link(solvers)
    capi.group_run(solvers, nsteps)
    for r, (b, s) in enumerate(zip(boxes, solvers)):
        tm1, tm2 = s.download()
        assert H.rel_linf(tm1, o2[b.gid]) < TOL, (r, "tm1")
        assert H.rel_linf(tm2, o1[b.gid]) < TOL, (r, "tm2")
        s.close()
    for b in boxes:
        b.close()
    # uniform box large enough for lattice patches (32 x 32 x 32: 8 interior patches of 64)
    nx = ny = nz = 32
    lnid, node_ijk, elem_ijk, et, nt = _box(nx, ny, nz, h=10.0, dt=2e-4)
    N = len(node_ijk)
    v1 = rng.uniform(-1, 1, (N, 3)) * 1e-3
    v2 = v1 + rng.uniform(-1, 1, (N, 3)) * 1e-6
    loaded = np.array([5, 4000, 20000], np.int32)
    F = rng.uniform(-1, 1, (6, 3, 3)) * 1e6
    p1, p2 = v2.copy(), v1.copy()
    ho.solver_run(lnid, et, nt, p1, p2, 0, 6, 2e-4, loaded_lnid=loaded, forces=F)
    s = ha.Solver(lnid, et, nt, 2e-4, tm1=v1, tm2=v2, node_xyz=_ticks(node_ijk, 1 << 20), variant=ha.HQ_VARIANT_PATCH)
    if brick_mode == "bricks":                      # the 31^3 simple nodes and the interior of the two z faces (31^2 each: the
                                                    # free surface and the bottom dashpot face ride with their columns) are
                                                    # brick nodes; the patch kernels keep the rest of the shell
        assert s.dominant_kernel() == "hq_k_brick" and s.info()["brick_nodes"] == 31 * 31 * 33
    elif ragged:                                    # all 64 patches are lattice subsets (dashpot faces included)
        assert s.dominant_kernel() == "hq_k_patch_stencil" and s.info()["ragged_patches"] > 0
        assert s.info()["stencil_patches"] == s.info()["npatches"]
    else:                                           # 8 stencil patches of 64: the element kernel is still the dominant one
        assert s.dominant_kernel() == want and s.info()["ragged_patches"] == 0
    s.set_source(loaded, F)
    s.run(6)
    tm1, tm2 = s.download()
    assert H.rel_linf(tm1, p2) < TOL and H.rel_linf(tm2, p1) < TOL
    s.close()
