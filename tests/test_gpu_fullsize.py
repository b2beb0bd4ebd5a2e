"""Parity at BASELINE.json's full sizes through size-independent properties
(the oracle cannot finish 8M / 64M elements in seconds):

* the two independent kernel paths (atomic scatter vs fused patches) agree;
* the step is linear in (tm1, tm2, source): step(a u + b v) = a step(u) + b step(v);
* a quiescent field stays exactly zero; all values stay finite;
* at 1M elements the oracle itself is still affordable for a few steps.
All through the C-ABI, meshes built by the C host side."""
import numpy as np
import pytest

import hercules_amd as ha
from hercules_amd import host
from oracle import herc_oracle as ho
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _field(box, seed, amp=1e-3):
    ijk = box.node_ijk.astype(np.int64)
    gid = (ijk[:, 2] * (box.ny + 1) + ijk[:, 1]) * (box.nx + 1) + ijk[:, 0]
    u = np.empty((len(gid), 3))
    for d in range(3):
        x = (gid * 3 + d + seed) * np.int64(2654435761) % np.int64(2 ** 31)
        u[:, d] = (x.astype(np.float64) / 2 ** 30 - 1.0) * amp
    return u


def _run(box, variant, u1, u2, nsteps, src=None, options=None, want=None):
    s = box.create_solver(variant=variant, tm1=u1, tm2=u2, options=options)
    if want:
        info = s.info()
        assert all(want[k](info) for k in want), info
    if src is not None:
        s.set_source(src[0], src[1])
    s.run(nsteps)
    out = s.download()
    s.close()
    return out


@pytest.mark.parametrize("wl", ["c2", "c2-nobricks"])     # (the 64M box: 75 oracle cone windows below, and 8 partitions against one)
def test_fullsize_variants_agree_and_step_is_linear(wl, monkeypatch):
    if wl.endswith("-nobricks"):     # the patch kernels alone (lattice / ragged stencil patches, element form), as without node_xyz
        monkeypatch.setenv("HQ_NO_BRICKS", "1")
        wl = wl.split("-")[0]
    nx, ny, nz, h, dt, freq = {"c2": (256, 256, 128, 1000.0 / 256, 1.8e-4, 100.0),
                               "c3": (512, 512, 256, 1000.0 / 512, 9.0e-5, 200.0)}[wl]
    box = host.Box(nx, ny, nz, h, dt, freq)
    assert box.info["total_elements"] == {"c2": 8388608, "c3": 67108864}[wl]
    nsteps = 4
    L = nx * h
    loaded, pattern = box.point_source(L / 2, L / 2, L / 5, 0.0, 90.0, 0.0)
    rp = box.run_params(loaded=loaded, pattern=pattern, moment=1e12, rise_time=3 * dt)
    F = box.source_table(rp, 0, nsteps)
    u = _field(box, 12345)
    v = _field(box, 777, 2e-3)
    a, b = 0.75, -1.5
    pu = _run(box, ha.HQ_VARIANT_PATCH, u, 0.999 * u, nsteps, (loaded, F))
    su = _run(box, ha.HQ_VARIANT_SCATTER, u, 0.999 * u, nsteps, (loaded, F))
    scale = np.abs(pu[0]).max()
    assert np.isfinite(pu[0]).all() and scale > 0
    assert np.abs(pu[0] - su[0]).max() <= 1e-11 * scale          # two independent kernel paths
    assert np.abs(pu[1] - su[1]).max() <= 1e-11 * scale
    del su
    pv = _run(box, ha.HQ_VARIANT_PATCH, v, 1.001 * v, nsteps, (loaded, 2.0 * F))
    w1 = a * u + b * v
    w2 = a * 0.999 * u + b * 1.001 * v
    pw = _run(box, ha.HQ_VARIANT_PATCH, w1, w2, nsteps, (loaded, (a + 2.0 * b) * F))
    lin = a * pu[0] + b * pv[0]
    assert np.abs(pw[0] - lin).max() <= 1e-11 * np.abs(lin).max()
    if wl != "c3":                                       # (quiescence at 8M; the 64M box has its cone windows)
        z = _run(box, ha.HQ_VARIANT_PATCH, np.zeros_like(u), np.zeros_like(u), nsteps)
        assert not z[0].any() and not z[1].any()
    box.close()


def test_one_million_elements_against_oracle():
    nx, ny, nz, h, dt, freq = 128, 128, 64, 1000.0 / 128, 3.6e-4, 50.0
    box = host.Box(nx, ny, nz, h, dt, freq)
    u = _field(box, 4242)
    nsteps = 3
    o1, o2 = (0.999 * u).copy(), u.copy()                            # oracle arrays are pre-swap
    ho.solver_run(box.lnid, box.etable.copy(), box.ntable.copy(), o1, o2, 0, nsteps, dt)
    for variant in (ha.HQ_VARIANT_PATCH, ha.HQ_VARIANT_SCATTER):
        tm1, tm2 = _run(box, variant, u, 0.999 * u, nsteps)
        assert H.rel_linf(tm1, o2) < 1e-9 and H.rel_linf(tm2, o1) < 1e-9
    box.close()


def test_one_million_elements_with_lateral_material_against_oracle():
    """bench.py's m1h (the 1 M-element cut of c3h): every element column has its own Vp, Vs, rho, so every element its
    own (c1, c2, beta) and every node its own n_t row -- what solver_init builds on a real CVM mesh
    (psolve.c:3360-3473).  No uniform-coefficient fast path applies; the oracle's loops on the same tables are the
    reference.  Also: two block partitions of it against the whole."""
    import bench
    from hercules_amd import capi
    nx, ny, nz, h, dt, freq = bench.WORKLOADS["m1h"]
    ncls, amp = bench.LATERAL["m1h"]
    box = host.Box(nx, ny, nz, h, dt, freq, lateral_classes=ncls, lateral_amp=amp)
    assert len(np.unique(box.etable[:, 0])) > 50
    u = _field(box, 777)
    nsteps = 3
    o1, o2 = (0.999 * u).copy(), u.copy()
    ho.solver_run(box.lnid, box.etable.copy(), box.ntable.copy(), o1, o2, 0, nsteps, dt)
    for variant in (ha.HQ_VARIANT_PATCH, ha.HQ_VARIANT_SCATTER):
        tm1, tm2 = _run(box, variant, u, 0.999 * u, nsteps)
        assert H.rel_linf(tm1, o2) < 1e-9 and H.rel_linf(tm2, o1) < 1e-9
    # the box hands over hq_desc.edata: every HET unit travels packed (12 B per element: rho, Vs, Vp; 16 B per node) ...
    packed = _run(box, ha.HQ_VARIANT_PATCH, u, 0.999 * u, nsteps,
                  want={"packed": lambda i: i["brick_units_packed"] == i["brick_units_het"] > 0})
    # ... and with hq_options.brick_no_pack = 1 as the 24-byte (c1, c2, beta) + 24-byte n_t rows of the rounds before
    plain = _run(box, ha.HQ_VARIANT_PATCH, u, 0.999 * u, nsteps, options={"brick_no_pack": 1},
                 want={"plain": lambda i: i["brick_units_packed"] == 0 and i["brick_units_het"] > 0})
    assert H.rel_linf(plain[0], o2) < 1e-9 and H.rel_linf(packed[0], o2) < 1e-9
    # the coefficients are the caller's doubles bit for bit either way; the n_t rows differ by <= 1e-15 (m2 = 2 m0 - (m0 - m1))
    assert H.rel_linf(packed[0], plain[0]) < 1e-12
    gid = (box.node_ijk[:, 2].astype(np.int64) * (ny + 1) + box.node_ijk[:, 1]) * (nx + 1) + box.node_ijk[:, 0]
    lut = np.empty(gid.max() + 1, np.int64)
    lut[gid] = np.arange(len(gid))
    box.close()
    parts = [host.Box(nx, ny, nz, h, dt, freq, lateral_classes=ncls, lateral_amp=amp, rank=r, nranks=2) for r in range(2)]
    maps = [lut[(b.node_ijk[:, 2].astype(np.int64) * (ny + 1) + b.node_ijk[:, 1]) * (nx + 1) + b.node_ijk[:, 0]] for b in parts]
    solvers = [b.create_solver(tm1=u[m], tm2=0.999 * u[m]) for b, m in zip(parts, maps)]
    capi.group_link(solvers)
    capi.group_run(solvers, nsteps)
    for s, m in zip(solvers, maps):
        tm1, tm2 = s.download()
        assert H.rel_linf(tm1, o2[m]) < 1e-9 and H.rel_linf(tm2, o1[m]) < 1e-9
        s.close()
    for b in parts:
        b.close()


def test_packed_n_t_rows_over_ten_thousand_steps():
    """The packed per-element units carry n_t as {m0, m0 - m1}: m1 comes out exactly, m2 = 2 m0 - (m0 - m1) equals the
    caller's to 1e-15 relative (checked per node at hq_create) -- the class of the summation-order difference, but a
    difference in EVERY step (round-5 advisor).  10 000 steps of a 64 x 64 x 32 box whose material differs from element
    to element, with a point source, packed against hq_options.brick_no_pack = 1 (the caller's 24-byte rows): the two fields
    stay within 1e-10 of the field's scale -- no drift beyond what the rounding of a step allows (measured: ~1e-13)."""
    nx, ny, nz, h, dt, freq = 64, 64, 32, 15.0, 3e-4, 30.0
    box = host.Box(nx, ny, nz, h, dt, freq, lateral_classes=61, lateral_amp=0.1)
    u = _field(box, 4242)
    nsteps = 10000
    L = nx * h
    loaded, pattern = box.point_source(L / 2 + 3.0, L / 2 - 2.0, L / 5, 30.0, 70.0, 10.0)
    rp = box.run_params(loaded=loaded, pattern=pattern, moment=1e13, rise_time=20 * dt, source_window=400)
    F = box.source_table(rp, 0, 400)
    out = []
    for no_pack in (0, 1):
        s = box.create_solver(tm1=u, tm2=0.999 * u, options={"brick_no_pack": no_pack})
        info = s.info()
        assert info["brick_units_het"] > 0 and (info["brick_units_packed"] == 0) == bool(no_pack)
        s.set_source(loaded, F)
        s.run(nsteps)
        out.append(s.download())
        assert s.check_finite() == 0
        s.close()
    box.close()
    scale = max(np.abs(out[1][0]).max(), 1e-300)
    assert scale > 0
    assert np.abs(out[0][0] - out[1][0]).max() <= 1e-10 * scale and np.abs(out[0][1] - out[1][1]).max() <= 1e-10 * scale


def test_host_solver_run_with_stations():
    """hqh_solver_run (the C mirror of solver_run): source windows, station
    cadence and interpolation, against the oracle driven the same way."""
    nx, ny, nz, h, dt, freq = 16, 16, 8, 62.5, 1e-3, 5.0
    box = host.Box(nx, ny, nz, h, dt, freq)
    loaded, pattern = box.point_source(500.0, 500.0, 100.0, 0.0, 90.0, 0.0)
    ids, phi, mine = box.stations(H.C1_STATIONS)
    got = {}
    rp = box.run_params(loaded=loaded, pattern=pattern, moment=1e15, rise_time=0.05, source_window=37,
                        station_ids=ids, station_phi=phi, station_rate=5,
                        station_fn=lambda step, disp: got.__setitem__(step, disp))
    nsteps = 200
    s = box.create_solver()
    box.solver_run(s, rp, 0, nsteps)
    F = box.source_table(rp, 0, nsteps)
    o1, o2 = np.zeros((box.info["nharbored"], 3)), np.zeros((box.info["nharbored"], 3))
    cap = ho.solver_run(box.lnid, box.etable.copy(), box.ntable.copy(), o1, o2, 0, nsteps, dt,
                        loaded_lnid=loaded, forces=F, cap_lnid=ids)
    st = np.einsum("sn,tsnd->tsd", phi, cap.reshape(nsteps, len(phi), 8, 3))
    assert sorted(got) == list(range(0, nsteps, 5))
    scale = np.abs(st).max()
    for step, disp in got.items():
        assert np.abs(disp - st[step]).max() <= 1e-9 * scale
    tm1, tm2 = s.download()
    assert H.rel_linf(tm1, o2) < 1e-9
    s.close()
    box.close()


def test_all_c_host_program(tmp_path):
    """examples/hq_psolve_mini.c: a host written entirely in C on the two C-ABI libraries
    (mesh + solver_init, point source, stations, solver_run, checkpoint) -- its station
    files and checkpoint against the oracle driven the same way."""
    import os
    import subprocess
    from hercules_amd import build as hbuild
    exe = hbuild.build_example()
    nx, ny, nz, h, dt, freq, nsteps = 16, 16, 8, 62.5, 1e-3, 5.0, 300
    out = subprocess.run([exe, str(nx), str(ny), str(nz), str(h), str(dt), str(freq), str(nsteps), str(tmp_path)],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=300)
    assert out.returncode == 0, out.stdout
    assert "Total elements: 2048" in out.stdout and "steps run: %d" % nsteps in out.stdout
    # the same problem through the oracle
    box = host.Box(nx, ny, nz, h, dt, freq)
    L, Lz = nx * h, nz * h
    loaded, pattern = box.point_source(L / 2, L / 2, Lz / 5, 0.0, 90.0, 0.0)
    pts = [(L / 2, L / 2, 0.0), (0.6 * L, 0.6 * L, 0.0), (0.75 * L, 0.75 * L, Lz / 4)]
    ids, phi, _ = box.stations(pts)
    rp = box.run_params(loaded=loaded, pattern=pattern, moment=1e15, rise_time=40 * dt)
    F = box.source_table(rp, 0, nsteps)
    N = box.info["nharbored"]
    o1, o2 = np.zeros((N, 3)), np.zeros((N, 3))
    cap = ho.solver_run(box.lnid, box.etable.copy(), box.ntable.copy(), o1, o2, 0, nsteps, dt,
                        loaded_lnid=loaded, forces=F, cap_lnid=ids)
    st = np.einsum("sn,tsnd->tsd", phi, cap.reshape(nsteps, 3, 8, 3))
    scale = np.abs(st).max()
    for s in range(3):
        rows = [l.split() for l in open(os.path.join(str(tmp_path), "station.%d" % s)).read().splitlines()[1:]]
        got = np.array([[float(v) for v in r] for r in rows])
        assert len(got) == nsteps // 10
        assert np.allclose(got[:, 0], np.arange(0, nsteps, 10) * dt, atol=1e-9)
        assert np.abs(got[:, 1:] - st[::10, s, :]).max() <= 6e-7 * scale     # "% 8e" text precision
    b = open(os.path.join(str(tmp_path), "checkpoint.out0"), "rb").read()
    assert list(np.frombuffer(b[:12], "<i4")) == [1, nsteps, N]
    tm2 = np.frombuffer(b[12:12 + N * 24], "<f8").reshape(N, 3)
    tm1 = np.frombuffer(b[12 + N * 24:], "<f8").reshape(N, 3)
    assert H.rel_linf(tm1, o2) < 1e-9 and H.rel_linf(tm2, o1) < 1e-9
    box.close()


_WHOLE = {}     # the 64M box and ONE solver context on it, shared by the two tests that need the whole box (4 s of mesh and
                # 8 s of hq_create each): test_eight_partitions...[c3] and test_dependency_cone_windows...[c3], which
                # follows it in this file and releases both


def _whole_box(wl):
    """-> (Box, Solver) of a uniform box workload, built once; the solver is handed over with whatever state it has --
    the caller uploads its start field and sets its source."""
    import bench
    if wl not in _WHOLE:
        nx, ny, nz, h, dt, freq = bench.WORKLOADS[wl]
        box = host.Box(nx, ny, nz, h, dt, freq)
        _WHOLE[wl] = (box, box.create_solver())
    return _WHOLE[wl]


def _release_whole_box(wl):
    if wl in _WHOLE:
        box, s = _WHOLE.pop(wl)
        s.close()
        box.close()


@pytest.mark.parametrize("wl,overlap,ragged,bricks", [("m1", 1, 1, 0), ("m1", 0, 1, 1), ("c3", 1, 1, 1),
                                                      ("m1", 0, 1, 2), ("c2h", 1, 1, 1)])
def test_eight_partitions_of_the_8m_box_match_one_partition(wl, overlap, ragged, bricks, monkeypatch):
    """BASELINE config 4 on one GPU: the 8M and the 64M box cut 8 ways (octor blocks),
    stepped with the in-process transport and the comm/compute overlap, against the
    single-partition run (the switches that differ only in a launch or transport detail on the 1M box; c2h: the 8M box
    with material of its own in every element, the het kernel on every partition and at every interface).  ragged: the lattice-subset patches (domain faces, partition interfaces) through
    hq_k_patch_stencil as well (HQ_PATCH_RAGGED=1: its launch ahead of the exchange, forces handed to the interface)."""
    from hercules_amd import capi
    monkeypatch.setenv("HQ_OVERLAP", str(overlap))
    monkeypatch.setenv("HQ_PATCH_RAGGED", "1" if ragged else "0")      # (the default is 1)
    if not bricks:                                                      # the patch kernels take every node
        monkeypatch.setenv("HQ_NO_BRICKS", "1")
    if bricks == 2:                                                     # the in-process transport with copies instead of
        monkeypatch.setenv("HQ_GROUP_COPIES", "1")                      # the pack kernel writing into the peers' buffers
    import bench
    nx, ny, nz, h, dt, freq = bench.WORKLOADS[wl]
    ncls, amp = bench.LATERAL.get(wl, (0, 0.0))        # c2h: material of its own in every element (hq_k_brick_het in every partition)
    nsteps = 3
    lap = H.lap_timer(wl + " x 8")
    shared = wl == "c3"                                  # (the default build of the 64M box: the cone-window test reuses it)
    if shared:
        one, s_one = _whole_box(wl)
        u = _field(one, 31337)
        lap("whole box + field")
        s_one.set_source(np.zeros(0, np.int32), np.zeros((0, 0, 3)))
        s_one.upload(u, 0.999 * u, 0)
        s_one.run(nsteps)
        ref1, ref2 = s_one.download()
    else:
        one = host.Box(nx, ny, nz, h, dt, freq, lateral_classes=ncls, lateral_amp=amp)
        u = _field(one, 31337)
        lap("whole box + field")
        ref1, ref2 = _run(one, ha.HQ_VARIANT_PATCH, u, 0.999 * u, nsteps)
    lap("whole box: hq_create, run, download")
    gid_one = (one.node_ijk[:, 2].astype(np.int64) * (ny + 1) + one.node_ijk[:, 1]) * (nx + 1) + one.node_ijk[:, 0]
    lut = np.empty(gid_one.max() + 1, np.int64)
    lut[gid_one] = np.arange(len(gid_one))
    if not shared:
        one.close()
    # (the ranks' meshes side by side: the C host side releases the GIL)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(8) as pool:
        boxes = list(pool.map(lambda r: host.Box(nx, ny, nz, h, dt, freq, rank=r, nranks=8, lateral_classes=ncls, lateral_amp=amp), range(8)))
    lap("8 rank boxes")
    # ... and so are the eight hq_create calls (thread-safe by design: errors and options in force are thread-local)
    def make(b):
        g = (b.node_ijk[:, 2].astype(np.int64) * (ny + 1) + b.node_ijk[:, 1]) * (nx + 1) + b.node_ijk[:, 0]
        m = lut[g]
        return m, b.create_solver(tm1=u[m], tm2=0.999 * u[m])
    with ThreadPoolExecutor(8) as pool:
        made = list(pool.map(make, boxes))
    solvers, maps = [], []
    for m, sv in made:
        maps.append(m)
        solvers.append(sv)
        assert solvers[-1].info()["variant"] == ha.HQ_VARIANT_PATCH
        if bricks:
            assert solvers[-1].dominant_kernel() == "hq_k_brick"
            if ncls:
                assert solvers[-1].info()["brick_units_het"] == solvers[-1].info()["brick_units"] > 0
        else:
            assert (solvers[-1].info()["ragged_patches"] > 0) == bool(ragged)
    lap("8 x hq_create")
    capi.group_link(solvers)
    capi.group_run(solvers, nsteps)
    scale = np.abs(ref1).max()
    for s, m in zip(solvers, maps):
        tm1, tm2 = s.download()
        assert np.abs(tm1 - ref1[m]).max() <= 1e-11 * scale
        assert np.abs(tm2 - ref2[m]).max() <= 1e-11 * scale
        s.close()
    for b in boxes:
        b.close()
    lap("run, download, compare")


def _cone_windows(nx, ny, nz, src_elem):
    """Centres (element indices) of the windows: corners, edges, faces (all five dashpot faces and the free surface
    z = 0), the far-face cubes, the source element, the borders of brick tiles / chunks and of the 8^3 patches, and
    seeded interior points."""
    c = []
    for i in (0, nx - 1):
        for j in (0, ny - 1):
            for k in (0, nz - 1):
                c.append((i, j, k))                                                  # 8 corners
    mid = (nx // 2 + 3, ny // 2 - 5, nz // 2 + 1)
    for a in range(3):                                                               # 12 edges, 6 + 6 face points
        for s1 in (0, 1):
            for s2 in (0, 1):
                e = list(mid)
                o = [d for d in range(3) if d != a]
                e[o[0]] = 0 if s1 == 0 else (nx, ny, nz)[o[0]] - 1
                e[o[1]] = 0 if s2 == 0 else (nx, ny, nz)[o[1]] - 1
                c.append(tuple(e))
        for s in (0, 1):
            e = list(mid)
            e[a] = 0 if s == 0 else (nx, ny, nz)[a] - 1
            c.append(tuple(e))
            e2 = [(nx, ny, nz)[d] // 3 + 2 for d in range(3)]
            e2[a] = 1 if s == 0 else (nx, ny, nz)[a] - 2
            c.append(tuple(e2))
    c.append(tuple(src_elem))
    # brick tiles start at node 1 and are 64 x 8 nodes, chunks 32 planes; round-2 patches are 8^3 node cubes
    for i in (63, 64, 65, 128):
        for j in (7, 8, 9, 16):
            c.append((i, j, nz // 2 + 2))
    for k in (31, 32, 33, 64):
        c.append((nx // 2 - 7, ny // 2 + 9, k))
        c.append((64, 8, k))
    rng = np.random.default_rng(99)
    for _ in range(16):
        c.append((int(rng.integers(0, nx)), int(rng.integers(0, ny)), int(rng.integers(0, nz))))
    return c


@pytest.mark.parametrize("wl", ["c2", "c3", "c3h"])
def test_dependency_cone_windows_against_the_oracle(wl):
    """Oracle parity AT BASELINE sizes: after k steps a node depends on its k-ring only, so a window of the full box
    -- a block of 4^3 elements and k + 1 more layers around it, with the true eTable / nTable rows and the true start
    field -- stepped by the oracle's reference loops gives the exact values of the nodes that lie at least k layers
    inside every CUT face of the window (domain faces are no cuts).  >= 64 windows of the 8 M and the 64 M box (all
    five dashpot faces, the free surface, edges, corners, the far-face cubes, the source element, tile / chunk /
    patch borders, seeded interior points) against the GPU's whole-box result: <= 1e-9 of the field's scale.
    c3h: the 64 M box whose material differs from element to element (every interior node in hq_k_brick_het, 62 x 7-node
    tiles: the windows at i = 63..65, 128 and j = 7..9, 16 straddle their borders as well)."""
    import bench
    nx, ny, nz, h, dt, freq = bench.WORKLOADS[wl]
    k = 4
    ncls, amp = bench.LATERAL.get(wl, (0, 0.0))
    shared = wl == "c3"
    box = _whole_box(wl)[0] if shared else host.Box(nx, ny, nz, h, dt, freq, lateral_classes=ncls, lateral_amp=amp)
    u = _field(box, 2718)
    L = nx * h
    loaded, pattern = box.point_source(L / 2, L / 2, L / 5, 0.0, 90.0, 0.0)
    rp = box.run_params(loaded=loaded, pattern=pattern, moment=1e12, rise_time=20 * dt, source_window=k)
    F = box.source_table(rp, 0, k)
    if shared:
        s = _whole_box(wl)[1]
        s.upload(u, 0.999 * u, 0)
    else:
        s = box.create_solver(tm1=u, tm2=0.999 * u)
    assert s.dominant_kernel() == "hq_k_brick"
    if ncls:
        rep = box.brick_plan_check() if nx <= 256 else None      # (the 64 M plan is checked by the run itself)
        assert s.info()["brick_nodes"] > 0.97 * len(u) and (rep is None or rep["het_units"] == rep["units"])
    s.set_source(loaded, F)
    s.run(k)
    ijk = box.node_ijk
    gid = (ijk[:, 2].astype(np.int64) * (ny + 1) + ijk[:, 1]) * (nx + 1) + ijk[:, 0]
    lut = np.empty((nx + 1) * (ny + 1) * (nz + 1), np.int32)
    lut[gid] = np.arange(len(gid), dtype=np.int32)
    e_of = np.full(len(gid), -1, np.int32)                    # element whose corner 0 a node is
    e_of[box.lnid[:, 0]] = np.arange(len(box.lnid), dtype=np.int32)
    src_elem = ijk[loaded].min(axis=0)
    scale = np.abs(u).max()
    dims = (nx, ny, nz)
    worst, nwin, nchecked = 0.0, 0, 0
    for ctr in _cone_windows(nx, ny, nz, src_elem):
        lo = [max(0, min(ctr[d], dims[d] - 4) - (k + 1)) for d in range(3)]
        hi = [min(dims[d], lo[d] + 4 + 2 * (k + 1)) for d in range(3)]
        ei, ej, ek = np.meshgrid(np.arange(lo[0], hi[0]), np.arange(lo[1], hi[1]), np.arange(lo[2], hi[2]), indexing="ij")
        e = e_of[lut[(ek.ravel().astype(np.int64) * (ny + 1) + ej.ravel()) * (nx + 1) + ei.ravel()]]
        assert (e >= 0).all()
        nodes, inv = np.unique(box.lnid[e], return_inverse=True)
        lnid_w = inv.reshape(-1, 8).astype(np.int32)
        o1, o2 = (0.999 * u[nodes]).copy(), u[nodes].copy()
        pos = {int(n): i for i, n in enumerate(nodes)}
        lw = [(pos[int(n)], i) for i, n in enumerate(loaded) if int(n) in pos]
        kw = {}
        if lw:
            kw = dict(loaded_lnid=np.array([a for a, _ in lw], np.int32),
                      forces=np.ascontiguousarray(F[:, [b for _, b in lw], :]))
        ho.solver_run(lnid_w, box.etable[e].copy(), box.ntable[nodes].copy(), o1, o2, 0, k, dt, **kw)
        q = ijk[nodes]
        ok = np.ones(len(nodes), bool)
        for d in range(3):
            if lo[d] > 0:
                ok &= q[:, d] >= lo[d] + k
            if hi[d] < dims[d]:
                ok &= q[:, d] <= hi[d] - k
        assert ok.sum() >= 27
        tm1, tm2 = s.gather(nodes[ok].astype(np.int32))
        worst = max(worst, np.abs(tm1 - o2[ok]).max() / scale, np.abs(tm2 - o1[ok]).max() / scale)
        nwin += 1
        nchecked += int(ok.sum())
    if shared:
        _release_whole_box(wl)
    else:
        s.close()
        box.close()
    assert nwin >= 64 and nchecked > 64 * 27
    assert worst < 1e-9, worst


def test_large_two_level_box_variants_agree():
    """A 6M-element two-level octree box from the C host (131k hanging nodes): the fused
    patch kernel (hanging-node accumulators + in-LDS distribution) against the scatter
    kernels + compute_adjust kernels, two independent implementations."""
    ob = host.OctBox(256, 256, 32, 48, 1000.0 / 256, 1.8e-4, 100.0)
    assert ob.ldnnum == 257 * 257 - 129 * 129
    rng = np.random.default_rng(5)
    u1 = rng.uniform(-1, 1, (ob.N, 3)) * 1e-3
    u2 = u1 * 0.999
    ho.compute_adjust(u1, 1, ob.dangling)
    ho.compute_adjust(u2, 1, ob.dangling)
    res = []
    for variant in (ha.HQ_VARIANT_PATCH, ha.HQ_VARIANT_SCATTER):
        s = ob.create_solver(variant=variant, tm1=u1, tm2=u2)
        assert s.info()["variant"] == variant
        s.run(5)
        res.append(s.download())
        s.close()
    scale = np.abs(res[0][0]).max()
    assert np.isfinite(scale) and scale > 0
    assert np.abs(res[0][0] - res[1][0]).max() <= 1e-11 * scale
    assert np.abs(res[0][1] - res[1][1]).max() <= 1e-11 * scale
    ob.close()


# ---------------------------------------------------------------------------------------------
# BASELINE config 5: the layered basin on four octree levels (bench.py workloads o3s / o3)
# ---------------------------------------------------------------------------------------------

def _basin(workload, rank=0, nranks=1, want_interfaces=False):
    """(OctBox, total elements, start field of its harbored nodes) of bench.py's layered-basin workloads."""
    import bench
    box, total_e, total_n, interfaces = bench.make_octbox(workload, rank, nranks)
    nx, ny = bench.WORKLOADS[workload][:2]
    out = (box, total_e, total_n, bench.seeded_field(box.node_xyz, nx, ny, interfaces))
    return out + (interfaces,) if want_interfaces else out


def _run_oct(box, variant, u1, u2, nsteps):
    s = box.create_solver(variant=variant, tm1=u1, tm2=u2)
    assert s.info()["variant"] == variant
    s.run(nsteps)
    out = s.download()
    s.close()
    return out


def test_small_basin_variants_agree_and_step_is_linear():
    """o3s (3.3M elements on four octree levels, hanging nodes on three interfaces): fused patch
    kernel vs scatter + compute_adjust kernels, linearity of the step, quiescence."""
    box, E, N, u = _basin("o3s")
    assert box.E == E and box.N == N and box.ldnnum > 0
    nsteps = 4
    v = np.roll(u, 1, axis=1) * 2.0                  # hanging rows stay means of their anchors (linear)
    a, b = 0.75, -1.5
    pu = _run_oct(box, ha.HQ_VARIANT_PATCH, u, 0.999 * u, nsteps)
    su = _run_oct(box, ha.HQ_VARIANT_SCATTER, u, 0.999 * u, nsteps)
    scale = np.abs(pu[0]).max()
    assert np.isfinite(pu[0]).all() and scale > 0
    assert np.abs(pu[0] - su[0]).max() <= 1e-11 * scale
    assert np.abs(pu[1] - su[1]).max() <= 1e-11 * scale
    pv = _run_oct(box, ha.HQ_VARIANT_PATCH, v, 1.001 * v, nsteps)
    pw = _run_oct(box, ha.HQ_VARIANT_PATCH, a * u + b * v, a * 0.999 * u + b * 1.001 * v, nsteps)
    lin = a * pu[0] + b * pv[0]
    assert np.abs(pw[0] - lin).max() <= 1e-11 * np.abs(lin).max()
    z = _run_oct(box, ha.HQ_VARIANT_PATCH, np.zeros_like(u), np.zeros_like(u), nsteps)
    assert not z[0].any() and not z[1].any()
    # hanging nodes carry the mean of their anchors after every step (compute_adjust ASSIGNMENT, psolve.c:5992-6035)
    chk = pu[0].copy()
    ho.compute_adjust(chk, 1, box.dangling)
    assert np.abs(chk - pu[0]).max() <= 1e-13 * scale
    box.close()


@pytest.mark.parametrize("variant,overlap", [(ha.HQ_VARIANT_PATCH, 1)])      # (round 6: the patch variant with the chain on its own stream;
                                                                             # scatter kernels on partitions: tests/test_gpu_parity.py)
def test_small_basin_on_eight_partitions_matches_one_partition(variant, overlap, monkeypatch):
    """o3s cut into octor's 8 partitions, every rank's tables built by that rank alone (octbox_local: hanging nodes and
    their anchors on partition interfaces, all four exchanges of a step, psolve.c:4298-4315), in-process transport, against
    the whole basin on one partition.  (The 189M-element basin itself in 8 partitions: the next test but one.)"""
    from hercules_amd import capi
    monkeypatch.setenv("HQ_OVERLAP", str(overlap))      # 1: the exchange chain beside the interior patches, as between GPUs
    nsteps = 5
    one, E, N, u = _basin("o3s")
    ref1, ref2 = _run_oct(one, ha.HQ_VARIANT_PATCH, u, 0.999 * u, nsteps)
    key = lambda xyz: (xyz[:, 2].astype(np.int64) << 42) | (xyz[:, 1].astype(np.int64) << 21) | xyz[:, 0].astype(np.int64)
    k1 = key(one.node_xyz)
    order = np.argsort(k1)
    one.close()
    boxes, solvers, maps = [], [], []
    for r in range(8):
        b, _, _, ur = _basin("o3s", r, 8)
        m = order[np.searchsorted(k1[order], key(b.node_xyz))]        # harbored node -> node of the whole basin
        assert np.array_equal(k1[m], key(b.node_xyz))
        assert np.array_equal(ur, u[m])                                 # the start field is a function of position
        boxes.append(b)
        maps.append(m)
        solvers.append(b.create_solver(variant=variant, tm1=ur, tm2=0.999 * ur))
    assert sum(b.E for b in boxes) == E
    assert sum(b.ldnnum for b in boxes) > 0
    capi.group_link(solvers)
    capi.group_run(solvers, nsteps)
    scale = np.abs(ref1).max()
    for s, m in zip(solvers, maps):
        tm1, tm2 = s.download()
        assert np.abs(tm1 - ref1[m]).max() <= 1e-11 * scale
        assert np.abs(tm2 - ref2[m]).max() <= 1e-11 * scale
        s.close()
    for b in boxes:
        b.close()


def _basin_windows(nx, ny, nzt, interfaces, k):
    """Windows of the layered basin that straddle each level interface (z0 = its plane, c = the coarser edge there, both
    in finest-element units): in the interior, at a domain side face and in a domain corner."""
    out = []
    for z0, hf in interfaces:
        c = 2 * hf
        margin = 2 * k * c
        W = 2 * margin + 2 * c
        for x0, y0 in (((nx // 2) // c * c - W // 2 // c * c, (ny // 3) // c * c), (0, (ny // 2) // c * c), (nx - W, ny - W)):
            lo = [max(0, x0), max(0, y0), max(0, z0 - margin - c)]
            hi = [min(nx, lo[0] + W), min(ny, lo[1] + W), min(nzt, z0 + margin + c)]
            out.append((lo, hi, margin))
    return out


def test_full_basin_against_the_oracle_and_in_eight_partitions():
    """o3 = BASELINE config 5 at scale on ONE GPU: 189M elements on four octree levels, 1.0M hanging nodes.
    * Finiteness, hanging nodes = mean of their anchors (the scatter kernels and quiescence are compared on o3s).
    * ORACLE parity at this size: dependency-cone windows that straddle each of the three level interfaces -- hanging
      nodes, their anchors, compute_adjust's distribution and assignment (psolve.c:5936-6039) inside the checked region
      -- in the interior, at a domain face and in a corner, stepped by the oracle's reference loops with the true table
      rows (tests/helpers.octree_window; the window logic itself is pinned on a whole-mesh oracle run in
      tests/test_octree_windows_cpu.py): <= 1e-9 of the field's scale.
    * The basin in 8 partitions, every rank's tables built by that rank alone (octbox_local), in-process transport: all
      four exchanges of a step on 3-7 neighbours each, against the single-partition run.
    Host memory is kept lean: one field at a time, newest displacement only."""
    import gc
    import bench
    from hercules_amd import capi
    psutil = pytest.importorskip("psutil")
    if psutil.virtual_memory().available < 90 * 2 ** 30:
        pytest.skip("needs ~80 GiB of host memory for the 189M-element mesh tables")
    lap = H.lap_timer("o3")
    box, E, N, u, interfaces = _basin("o3", want_interfaces=True)
    lap("mesh + field")
    assert E > 180e6 and box.ldnnum > 1e6 and len(interfaces) == 3
    nsteps = 2
    scale0 = np.abs(u).max()
    res = []
    s = box.create_solver(variant=ha.HQ_VARIANT_PATCH, tm1=u, tm2=0.999 * u)
    assert s.info()["variant"] == ha.HQ_VARIANT_PATCH and s.info()["brick_nodes"] > 0.9 * N
    s.run(nsteps)
    lap("hq_create + run")
    # oracle windows across the level interfaces
    xyz = box.node_xyz
    elem_lo = xyz[box.lnid[:, 0]].astype(np.int32)
    elem_edge = xyz[box.lnid[:, 1], 0] - elem_lo[:, 0]
    nx, ny = bench.WORKLOADS["o3"][:2]
    worst, nchecked, nhang = 0.0, 0, 0
    wins = _basin_windows(nx, ny, int(xyz[:, 2].max()), interfaces, nsteps)
    # (every interface in the interior; the first one also at a domain face and in a corner -- the windows of the
    #  laterally refined basin o4 below carry the rest of the load since round 5)
    for lo, hi, margin in [wins[0], wins[1], wins[3], wins[6]]:
        win = H.octree_window(box.lnid, xyz, box.dangling, elem_lo, elem_edge, lo, hi, margin)
        g1, g2 = H.octree_window_oracle(win, box.etable, box.ntable, u, 0.999, nsteps, box.dt)
        ok, nodes = win["ok"], win["nodes"]
        tm1, tm2 = s.gather(nodes[ok])
        worst = max(worst, np.abs(tm1 - g1[ok]).max() / scale0, np.abs(tm2 - g2[ok]).max() / scale0)
        nchecked += int(ok.sum())
        nhang += int(np.isin(nodes[ok], box.dangling[0]).sum())
    del elem_lo, elem_edge
    lap("oracle windows")
    assert nchecked > 400 and nhang > 20, (nchecked, nhang)
    assert worst < 1e-9, worst
    tm1, _ = s.download(want_tm2=False)
    s.close()
    res.append(tm1)
    gc.collect()
    scale = np.abs(res[0]).max()
    assert np.isfinite(scale) and scale > 0 and np.isfinite(res[0]).all()
    # (hanging nodes = mean of their anchors over the whole field: checked on o3s, o4s and o4; here on a sample)
    ids, ptr, anc = box.dangling
    pick = np.linspace(0, len(ids) - 1, 20000).astype(np.int64)
    for k in pick[::400]:
        assert np.abs(res[0][ids[k]] - res[0][anc[ptr[k]:ptr[k + 1]]].mean(axis=0)).max() <= 1e-13 * scale
    gc.collect()
    # eight partitions, each built by its rank alone, against the single run (res[0]); compared at the harbored nodes
    # of every rank through their coordinates
    # (a dense table over the node lattice instead of a sort of 190 M keys: 3.4 GB for half a minute less)
    far = box.node_xyz.max(axis=0).astype(np.int64)
    key = lambda xyz: (xyz[:, 2].astype(np.int64) * (far[1] + 1) + xyz[:, 1]) * (far[0] + 1) + xyz[:, 0]
    lut = np.full(int((far[0] + 1) * (far[1] + 1) * (far[2] + 1)), -1, np.int32)
    lut[key(box.node_xyz)] = np.arange(box.N, dtype=np.int32)
    box.close()
    gc.collect()
    lap("download, anchors, node table")
    solvers, maps = [], []
    # the ranks' tables are built side by side (the C host side releases the GIL; a rank alone takes ~10 s)
    from concurrent.futures import ThreadPoolExecutor
    def make(r):                      # mesh, map to the single run's nodes and hq_create of one rank
        b = bench.make_octbox("o3", r, 8)[0]
        m = lut[key(b.node_xyz)]
        ur = u[m]                     # (the start field is a function of the coordinates: the whole mesh's values there)
        sv = b.create_solver(tm1=ur, tm2=0.999 * ur)
        b.close()
        return m, sv
    with ThreadPoolExecutor(8) as pool:
        for m, sv in pool.map(make, range(8)):
            assert (m >= 0).all()
            maps.append(m)
            solvers.append(sv)
            assert solvers[-1].info()["brick_nodes"] > 0
    gc.collect()
    del lut, u
    lap("8 ranks: meshes, maps, hq_create")
    capi.group_link(solvers)
    capi.group_run(solvers, nsteps)
    for sv, m in zip(solvers, maps):
        tm1, _ = sv.download(want_tm2=False)
        assert np.abs(tm1 - res[0][m]).max() <= 1e-11 * scale
        sv.close()
    del res
    gc.collect()


@pytest.mark.parametrize("path", ["hq_k_brick", "hq_k_patch_stencil"])
def test_long_run_of_the_stencil_path_agrees_with_the_scatter_kernels(path, monkeypatch):
    """1 M-element box, point source, 1500 steps: hq_k_brick (the plane sums of the assembled stencil, marching along
    z; the patch kernels at the faces) and, with HQ_NO_BRICKS=1, hq_k_patch_stencil (+ the element kernel at the
    faces) against the scatter variant -- formulations of the same operator (assembled 27-point stencil vs element
    by element with atomics) -- stay together to rounding over a long run; all stay finite."""
    if path != "hq_k_brick":
        monkeypatch.setenv("HQ_NO_BRICKS", "1")
    nx, ny, nz, h, dt, freq = 128, 128, 64, 1000.0 / 128, 3.6e-4, 50.0
    box = host.Box(nx, ny, nz, h, dt, freq)
    L = nx * h
    loaded, pattern = box.point_source(L / 2, L / 2, L / 5, 0.0, 90.0, 0.0)
    nsteps = 1500
    rp = box.run_params(loaded=loaded, pattern=pattern, moment=1e12, rise_time=20 * dt, source_window=nsteps)
    F = box.source_table(rp, 0, nsteps)
    res = []
    for variant, kernel in ((ha.HQ_VARIANT_PATCH, path), (ha.HQ_VARIANT_SCATTER, "hq_k_element_scatter")):
        s = box.create_solver(variant=variant)
        assert s.dominant_kernel() == kernel
        s.set_source(loaded, F)
        s.run(nsteps)
        res.append(s.download()[0])
        assert s.check_finite() == 0
        s.close()
    scale = np.abs(res[1]).max()
    assert scale > 0 and np.abs(res[0] - res[1]).max() <= 1e-11 * scale
    box.close()


def test_small_basin_against_the_oracle():
    """BASELINE config 5 in small (o3s: 3.3 M elements on four octree levels, hanging nodes on three interfaces, three
    materials): the oracle's reference loops with compute_adjust on the WHOLE mesh for two steps against the default
    path (bricks in every level's uniform interior, patches with hanging-node accumulators around them)."""
    box, E, N, u = _basin("o3s")
    nsteps = 2
    o1, o2 = (0.999 * u).copy(), u.copy()
    ho.solver_run(box.lnid, box.etable.copy(), box.ntable.copy(), o1, o2, 0, nsteps, box.dt, dangling=box.dangling)
    s = box.create_solver(tm1=u, tm2=0.999 * u)
    assert s.info()["brick_nodes"] > 0.3 * N
    s.run(nsteps)
    tm1, tm2 = s.download()
    s.close()
    assert H.rel_linf(tm1, o2) < 1e-9 and H.rel_linf(tm2, o1) < 1e-9
    box.close()


# ---------------------------------------------------------------------------------------------
# BASELINE config 5 with LATERAL refinement (bench.py workloads o4s / o4): a sediment bowl in a layered half-space,
# meshed by hqh_octree_generate as the reference's mesher would (pinned on tests/golden/c5_basin) -- level interfaces
# with x-, y- and z-normal faces and staircase corners, hanging nodes of every orientation
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("mode", ["bricks", "full-tiles-only", "patches-only", "scatter", "brick-stream-timed"])
def test_small_lateral_basin_against_the_oracle(mode, monkeypatch):
    """o4s (0.93 M elements on four levels, 62 k hanging nodes of all six kinds): the oracle's reference loops with
    compute_adjust on the WHOLE mesh for three steps against the default path (bricks where a level's interior is
    uniform -- full tile columns, and beside the x- and y-normal level interfaces and the bowl's staircase the RAGGED ones,
    HQ_BK_RAGGED --, patches around them), the same with full tile columns only (hq_options.brick_ragged = 0), the patch
    kernels alone and the scatter kernels.  Forces on 3 000 nodes all over the mesh at every step: the ragged units' nodes
    among them (compute_addforce_s, psolve.c:5912-5928)."""
    import bench
    if mode == "patches-only":
        monkeypatch.setenv("HQ_NO_BRICKS", "1")
    box, E, N, it = bench.make_octbox("o4s", 0, 1)
    u = it["field"]
    deps, mask, dist = H.hanging_kinds(box.node_xyz, box.dangling)
    assert set(mask.tolist()) == {1, 2, 3, 4, 5, 6}
    nsteps = 3
    free = np.setdiff1d(np.arange(N, dtype=np.int64), box.dangling[0])
    loaded = free[np.linspace(0, len(free) - 1, 3000).astype(np.int64)].astype(np.int32)
    rng = np.random.default_rng(99)
    F = rng.uniform(-1.0, 1.0, (nsteps, len(loaded), 3)) * (1e-4 * np.abs(u).max() / box.dt ** 2) * box.ntable[loaded, 0][None, :, None]
    o1, o2 = (0.999 * u).copy(), u.copy()
    ho.solver_run(box.lnid, box.etable.copy(), box.ntable.copy(), o1, o2, 0, nsteps, box.dt, dangling=box.dangling,
                  loaded_lnid=loaded, forces=F)
    # brick-stream-timed (round 6): the arrangement the 100 M-element basins step in -- the shell's patches on the compute
    # stream BESIDE the bricks on a stream of their own (hq_options.brick_stream; by default only above 4 096 patches) --
    # through hq_run_timed, where the hanging-node assignment is held back until the step's bricks have ended: the next
    # step's bricks, which read hanging nodes as neighbours, must wait for it (a dependency round 5 did not record)
    s = box.create_solver(variant=ha.HQ_VARIANT_SCATTER if mode == "scatter" else ha.HQ_VARIANT_PATCH, tm1=u, tm2=0.999 * u,
                          options={"brick_ragged": 0} if mode == "full-tiles-only" else
                                  ({"brick_stream": 1} if mode == "brick-stream-timed" else None))
    info = s.info()
    if mode == "bricks":
        assert info["brick_nodes"] > 0.6 * N and info["brick_units_ragged"] > 0
    elif mode == "full-tiles-only":
        assert 0.3 * N < info["brick_nodes"] < 0.6 * N and info["brick_units_ragged"] == 0
    elif mode == "patches-only":
        assert info["brick_nodes"] == 0
    s.set_source(loaded, F)
    if mode == "brick-stream-timed":
        assert s.info()["brick_stream"] == 0          # (the stream is made at the first step)
        s.run_timed(nsteps)
        # (a timed batch samples every fourth step's phase events; hq_options.phase_timing = 1 records every step)
        assert s.info()["brick_stream"] == 1 and s.info()["timed_steps"] >= 1 and s.info()["t_interior_us"] > 0
    else:
        s.run(nsteps)
    tm1, tm2 = s.download()
    s.close()
    assert H.rel_linf(tm1, o2) < 1e-9 and H.rel_linf(tm2, o1) < 1e-9
    box.close()


def test_small_gradient_basin_against_the_oracle():
    """o4gs (round 6): the same small basin with a VELOCITY GRADIENT -- 6 257 distinct materials, no two neighbouring coarse
    elements alike (what setrec's 27-sample average gives on any real CVM, psolve.c:1307-1397) -- so the per-element kernels
    run inside every octree level: full 62 x 7 tiles of hq_k_brick_het<PACKED>, beside the level interfaces the RAGGED ones
    (hq_k_brick_het<PACKED, RAGGED>), element-form patches with hanging nodes around them.  The oracle's reference loops with
    compute_adjust on the WHOLE mesh for three steps, forces on 3 000 nodes at every step, against the default path and
    against the path without ragged per-element units (hq_options.brick_ragged_het = 0)."""
    import bench
    box, E, N, it = bench.make_octbox("o4gs", 0, 1)
    u = it["field"]
    assert len(np.unique(box.etable[:, :2], axis=0)) > 5000
    nsteps = 3
    free = np.setdiff1d(np.arange(N, dtype=np.int64), box.dangling[0])
    loaded = free[np.linspace(0, len(free) - 1, 3000).astype(np.int64)].astype(np.int32)
    rng = np.random.default_rng(77)
    F = rng.uniform(-1.0, 1.0, (nsteps, len(loaded), 3)) * (1e-4 * np.abs(u).max() / box.dt ** 2) * box.ntable[loaded, 0][None, :, None]
    o1, o2 = (0.999 * u).copy(), u.copy()
    ho.solver_run(box.lnid, box.etable.copy(), box.ntable.copy(), o1, o2, 0, nsteps, box.dt, dangling=box.dangling,
                  loaded_lnid=loaded, forces=F)
    for ragged_het in (1, 0):
        s = box.create_solver(variant=ha.HQ_VARIANT_PATCH, tm1=u, tm2=0.999 * u, options={"brick_ragged_het": ragged_het})
        info = s.info()
        assert info["brick_units_het"] > 0 and info["brick_units_packed"] == info["brick_units_het"]
        if ragged_het:
            assert info["brick_units_ragged_het"] > 100 and info["brick_nodes"] > 0.8 * N
        else:
            assert info["brick_units_ragged_het"] == 0 and info["brick_nodes"] < 0.5 * N
        s.set_source(loaded, F)
        s.run(nsteps)
        tm1, tm2 = s.download()
        s.close()
        assert H.rel_linf(tm1, o2) < 1e-9 and H.rel_linf(tm2, o1) < 1e-9
    box.close()


@pytest.mark.parametrize("wl,nranks,overlap", [("o4s", 8, 0), ("o4gs", 8, 1)])       # (5 ranks: the reference's own 5-rank run of its basin, tests/test_gpu_parity.py and test_gpu_multiprocess.py)
def test_small_lateral_basin_in_partitions_matches_one_partition(wl, nranks, overlap, monkeypatch):
    """(o4gs: the basin with a velocity gradient -- per-element kernels, the RAGGED per-element units among them, on
    partitions with the chain on its own stream.)
    o4s cut into octor's block partitions (hqh_mesh_from_leaves with rank / nranks: ownership by Z-order point
    location, anchors of shared hanging nodes across x- / y- / z-normal interfaces), patch variant with bricks, in-process
    transport, against the whole basin on one partition.  overlap: the exchange chain on its own stream beside the
    interior launches -- the 100-register forms of hq_k_brick, the ragged one among them."""
    import bench
    from hercules_amd import capi
    monkeypatch.setenv("HQ_OVERLAP", str(overlap))
    nsteps = 4
    one, E, N, it = bench.make_octbox(wl, 0, 1)
    u = it["field"]
    ref1, ref2 = _run_oct(one, ha.HQ_VARIANT_PATCH, u, 0.999 * u, nsteps)
    one.close()
    solvers, gids = [], []
    bench.make_octbox(wl, 0, nranks)[0].close()        # (fills bench's cache of the leaves and the whole-mesh field)

    def make(r):
        b, _, _, itr = bench.make_octbox(wl, r, nranks)
        assert np.array_equal(itr["field"], u[b.gid])
        sv = b.create_solver(variant=ha.HQ_VARIANT_PATCH, tm1=itr["field"], tm2=0.999 * itr["field"])
        gid = b.gid.copy()
        b.close()
        return sv, gid
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(nranks) as pool:
        for sv, gid in pool.map(make, range(nranks)):
            solvers.append(sv)
            gids.append(gid)
    assert sum(sv.info()["brick_units_ragged_het" if wl == "o4gs" else "brick_units_ragged"] for sv in solvers) > 0
    capi.group_link(solvers)
    capi.group_run(solvers, nsteps)
    scale = np.abs(ref1).max()
    for sv, gid in zip(solvers, gids):
        tm1, tm2 = sv.download()
        assert np.abs(tm1 - ref1[gid]).max() <= 1e-11 * scale and np.abs(tm2 - ref2[gid]).max() <= 1e-11 * scale
        sv.close()


def test_full_lateral_basin_against_the_oracle():
    """o4 = BASELINE config 5's mesh class at scale on ONE GPU: 101 M elements on four octree levels whose interfaces
    follow a sediment bowl (x-, y- and z-normal faces, staircase corners).  ORACLE parity through dependency-cone
    windows centred on hanging nodes of every orientation and level pair (tests/helpers.lateral_windows, pinned on a
    whole-mesh oracle run in tests/test_octree_windows_cpu.py), stepped by the oracle's reference loops with the true
    table rows: <= 1e-9 of the field's scale.  Also: finite, hanging nodes = mean of their anchors."""
    import gc
    import bench
    psutil = pytest.importorskip("psutil")
    if psutil.virtual_memory().available < 70 * 2 ** 30:
        pytest.skip("needs ~60 GiB of host memory for the 101M-element mesh tables")
    lap = H.lap_timer("o4")
    box, E, N, it = bench.make_octbox("o4", 0, 1)
    u = it["field"]
    lap("mesh + field")
    assert E > 100e6 and box.ldnnum > 3e5
    nsteps = 2
    scale0 = np.abs(u).max()
    s = box.create_solver(variant=ha.HQ_VARIANT_PATCH, tm1=u, tm2=0.999 * u)
    assert s.info()["brick_nodes"] > 0.8 * N
    s.run(nsteps)
    lap("hq_create + run")
    xyz = box.node_xyz
    elem_lo = xyz[box.lnid[:, 0]].astype(np.int32)
    elem_edge = xyz[box.lnid[:, 1], 0] - elem_lo[:, 0]
    deps, mask, dist = H.hanging_kinds(xyz, box.dangling)
    assert set(mask.tolist()) == {1, 2, 3, 4, 5, 6}
    wins = H.lateral_windows(xyz, box.dangling, elem_lo, elem_edge, nsteps, per_kind=1)
    worst, nchecked, nhang, kinds = 0.0, 0, 0, set()
    for lo, hi, margin, centre, cand in wins:
        win = H.octree_window(box.lnid, xyz, box.dangling, elem_lo, elem_edge, lo, hi, margin, cand)
        g1, g2 = H.octree_window_oracle(win, box.etable, box.ntable, u, 0.999, nsteps, box.dt)
        ok, nodes = win["ok"], win["nodes"]
        assert centre in nodes[ok]
        tm1, tm2 = s.gather(nodes[ok])
        worst = max(worst, np.abs(tm1 - g1[ok]).max() / scale0, np.abs(tm2 - g2[ok]).max() / scale0)
        nchecked += int(ok.sum())
        hang = np.isin(box.dangling[0], nodes[ok])
        nhang += int(hang.sum())
        kinds |= set(zip(mask[hang].tolist(), dist[hang].tolist()))
    del elem_lo, elem_edge
    lap("oracle windows")
    assert len(wins) >= 12 and nchecked > 5000 and nhang > 200
    assert {m for m, _ in kinds} == {1, 2, 3, 4, 5, 6}
    assert worst < 1e-9, worst
    tm1, _ = s.download(want_tm2=False)
    s.close()
    scale = np.abs(tm1).max()
    assert np.isfinite(tm1).all() and scale > 0
    chk = tm1.copy()
    ho.compute_adjust(chk, 1, box.dangling)
    assert np.abs(chk - tm1).max() <= 1e-13 * scale
    box.close()
    gc.collect()
