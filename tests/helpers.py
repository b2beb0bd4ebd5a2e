"""Shared set-up of the reference's examples/simple case (C1) from the golden
fixtures: mesh in octor order, eTable/nTable from the oracle's solver_init."""
import os

import numpy as np

from oracle import herc_oracle as ho

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# octor root: 1000 m x 1000 m x 500 m -> farendp = (2^30, 2^30, 2^29) ticks (octor.c:4120-4140)
C1_FAR_TICKS = (2 ** 30, 2 ** 30, 2 ** 29)
C1_NX, C1_NY, C1_NZ = 16, 16, 8
C1_H = 62.5
C1_STATIONS = [(500.0 + 100.0 * i, 500.0 + 100.0 * i, 100.0) for i in range(5)]


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def c1_mesh():
    g = load("c1_short")
    lnid, node_ijk, elem_ijk, edge = ho.mesh_from_elem_ticks(g["elem_ticks"], C1_FAR_TICKS)
    mat = g["mat_vs_vp_rho"]
    edata = np.empty((len(lnid), 4), np.float32)
    edata[:, 0] = np.float32(1000.0 / 2 ** 30 * edge)
    edata[:, 1] = mat[:, 1]
    edata[:, 2] = mat[:, 0]
    edata[:, 3] = mat[:, 2]
    return lnid, node_ijk, elem_ijk, edata


def c1_problem(damping="rayleigh", real=np.float64):
    """real: solver_float (psolve.h:60-64) -- float32 = the tables as the reference's -DSINGLE_PRECISION_SOLVER build sums them."""
    lnid, node_ijk, elem_ijk, edata = c1_mesh()
    face = ho.face_bits(elem_ijk, C1_NX, C1_NY, C1_NZ)
    etable, ntable = ho.solver_init(lnid, edata.copy(), face, len(node_ijk), 1e-3, 5.0,
                                    damping=ho.DAMPING_BY_NAME[damping], real=real)
    return dict(lnid=lnid, node_ijk=node_ijk, elem_ijk=elem_ijk, edata=edata, face=face,
                etable=etable, ntable=ntable, N=len(node_ijk), E=len(lnid), dt=1e-3,
                damping=ho.DAMPING_BY_NAME[damping])


def rel_linf(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300))


def c5_problem(name="c5_two_level", real=np.float64):
    """One of the reference's own octree meshes (c5_two_level: soft layer refined one level
    deeper, 800 hanging nodes; c5_three_level: three element sizes and three materials that
    take every branch of mu_and_lambda) rebuilt from its flat dump, with eTable / nTable as
    solver_init leaves them (incl. the hanging-node mass distribution, psolve.c:3498-3507)."""
    g = load(name)
    m = ho.octree_mesh_from_elem_ticks(g["elem_ticks"], C1_FAR_TICKS)
    E, N = len(m["lnid"]), len(m["node_q"])
    mat = g["mat_vs_vp_rho"]
    tick = 1000.0 / 2 ** 30
    edata = np.empty((E, 4), np.float32)
    edata[:, 0] = (tick * m["emin"] * m["elem_size"].astype(np.float64)).astype(np.float32)
    edata[:, 1], edata[:, 2], edata[:, 3] = mat[:, 1], mat[:, 0], mat[:, 2]
    etable, ntable = ho.solver_init(m["lnid"], edata, m["face"], N, 1e-3, float(g["freq"]), real=real)
    ho.compute_adjust(ntable, 0, m["dangling"])
    return dict(lnid=m["lnid"], node_q=m["node_q"], etable=etable, ntable=ntable, dangling=m["dangling"],
                N=N, E=E, dt=1e-3, emin=m["emin"], golden=g, elem_size=m["elem_size"], edata=edata, freq=float(g["freq"]))


def c5_material(p):
    """(bBase, threshold_damping, threshold_vpvs) that c5_problem's solver_init combined p["edata"] with: hq_desc.mat_*."""
    return (ho.setab(p["freq"], ho.DAMP_RAYLEIGH)[1], 0.05, 3.0)


def two_level_mesh(nx, ny, nz_fine, nz_coarse, soft=(3000.0, 1732.0, 2200.0), hard=(6000.0, 3464.0, 2700.0),
                   h_fine=31.25, dt=1e-3, freq=5.0):
    """A layered box meshed on two octree levels, in octor's conventions: the top
    nz_fine layers of fine elements (nx x ny, edge h) over nz_coarse layers of elements of
    edge 2h -- the shape the reference's mesher produced for tests/golden/c5_two_level
    (where this construction is pinned bit-for-bit).  nx, ny, nz_fine even.
    -> dict like c5_problem()."""
    assert nx % 2 == 0 and ny % 2 == 0 and nz_fine % 2 == 0
    corners = np.array([[(c >> 0) & 1, (c >> 1) & 1, (c >> 2) & 1] for c in range(8)], np.int64)
    fi, fj, fk = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz_fine), indexing="ij")
    fine_ll = np.stack([fi.ravel(), fj.ravel(), fk.ravel()], 1).astype(np.int64)
    ci, cj, ck = np.meshgrid(np.arange(nx // 2), np.arange(ny // 2), np.arange(nz_coarse), indexing="ij")
    coarse_ll = np.stack([2 * ci.ravel(), 2 * cj.ravel(), nz_fine + 2 * ck.ravel()], 1).astype(np.int64)
    ll = np.concatenate([fine_ll, coarse_ll])
    size = np.concatenate([np.ones(len(fine_ll), np.int64), 2 * np.ones(len(coarse_ll), np.int64)])
    order = np.argsort(ho.zvalue(ll[:, 0], ll[:, 1], ll[:, 2]), kind="stable")   # octree pre-order
    ll, size = ll[order], size[order]
    ticks = ll[:, None, :] + corners[None, :, :] * size[:, None, None]
    far = (nx, ny, nz_fine + 2 * nz_coarse)
    m = ho.octree_mesh_from_elem_ticks(ticks, far)
    E, N = len(m["lnid"]), len(m["node_q"])
    edata = np.empty((E, 4), np.float32)
    edata[:, 0] = (h_fine * m["elem_size"]).astype(np.float32)
    is_fine = m["elem_size"] == 1
    for col, (a, b) in enumerate(zip(soft, hard)):
        edata[:, 1 + col] = np.where(is_fine, a, b)
    etable, ntable = ho.solver_init(m["lnid"], edata, m["face"], N, dt, freq)
    ho.compute_adjust(ntable, 0, m["dangling"])
    return dict(lnid=m["lnid"], node_q=m["node_q"], etable=etable, ntable=ntable, dangling=m["dangling"],
                N=N, E=E, dt=dt, emin=1, mesh=m)


def c5_np8_problem(name="c5_two_level_np8", real=np.float64):
    """One of the reference's octree meshes on several MPI ranks (c5_two_level_np8; c5_basin_np8 /
    c5_basin_np5: the laterally refined basin): octor's per-rank tables restated
    from the global view (ho.octree_partition), per-rank eTable / nTable after the mass
    exchange, the reference's per-rank force files and checkpoint stripes."""
    g = load(name)
    base = load(str(g["base"]))
    nranks = int(g["nranks"])
    m = ho.octree_mesh_from_elem_ticks(base["elem_ticks"], C1_FAR_TICKS)
    far_q = [f // m["emin"] for f in C1_FAR_TICKS]
    parts = ho.octree_partition(m, nranks, far_q)
    mat = base["mat_vs_vp_rho"]
    tick = 1000.0 / 2 ** 30
    E = len(m["lnid"])
    edata = np.empty((E, 4), np.float32)
    edata[:, 0] = (tick * m["emin"] * m["elem_size"].astype(np.float64)).astype(np.float32)
    edata[:, 1], edata[:, 2], edata[:, 3] = mat[:, 1], mat[:, 0], mat[:, 2]
    eds = [np.ascontiguousarray(edata[p["elems"]]) for p in parts]
    fcs = [np.ascontiguousarray(m["face"][p["elems"]]) for p in parts]
    ets, nts = ho.multi_rank_init(parts, eds, fcs, 1e-3, float(base["freq"]), real=real)
    return dict(golden=g, base=base, mesh=m, parts=parts, ets=ets, nts=nts, dt=1e-3, nranks=nranks, edata=edata,
                loaded=[g["loaded_lnid_%d" % r] for r in range(nranks)],
                forces=[g["forces_%d" % r] for r in range(nranks)])


# the material databases tests/golden/make_golden.py wrote with oracle/make_cvm (level-4 octants of 62.5 m), as grids in the
# MESH's axes [k][y][x]: cvm_query(east = y, north = x) (psolve.c:1352), so make_cvm's octant index i runs along mesh y
CVM_MODELS = {
    "c5_two_level": dict(layers=[(0, 3000, 1732, 2200), (2, 6000, 3464, 2700)], vscut=500, freq=5.0),
    "c5_three_level": dict(layers=[(0, 1500, 150, 1800), (2, 2500, 2000, 2300), (4, 6000, 3464, 2700)], vscut=100, freq=0.25),
    "c5_layered": dict(layers=[(0, 800, 200, 1700), (1, 1500, 450, 2000), (3, 2600, 1200, 2300)], vscut=100, freq=0.5),
    "c5_basin": dict(background=(6000, 3464, 2700), vscut=100, freq=5.0,
                     regions=[("dip", 3.2, -0.3, -0.12, 3000, 1732, 2200), ("box", 12, 16, 9, 13, 0, 2, 1500, 866, 1800)]),
    # the same basin with a velocity gradient (make_cvm's `grad`): a material of its own in every database octant
    "c5_gradient": dict(background=(6000, 3464, 2700), vscut=100, freq=5.0,
                        regions=[("dip", 3.2, -0.3, -0.12, 3000, 1732, 2200), ("box", 12, 16, 9, 13, 0, 2, 1500, 866, 1800),
                                 ("grad", 0.10, -0.06, 0.12)]),
}


def cvm_grid(name, level=4):
    """-> vp, vs, rho [nz][ny][nx] float32 (mesh axes), cell edge in metres."""
    spec = CVM_MODELS[name]
    n, nz = 1 << level, 1 << (level - 1)
    k, i, j = np.meshgrid(np.arange(nz), np.arange(n), np.arange(n), indexing="ij")     # [k][mesh y = i][mesh x = j]
    out = [np.zeros((nz, n, n), np.float32) for _ in range(3)]
    if "layers" in spec:
        for k0, vp, vs, rho in spec["layers"]:
            for a, v in zip(out, (vp, vs, rho)):
                a[k >= k0] = v
    else:
        for a, v in zip(out, spec["background"]):
            a[:] = v
        grade = None
        for r in spec["regions"]:
            if r[0] == "grad":               # make_cvm.c: evaluated in double, the products rounded to float
                grade = 1.0 + r[1] * (i + 0.5) / n + r[2] * (j + 0.5) / n + r[3] * (k + 0.5) / nz
                continue
            if r[0] == "box":
                sel = (i >= r[1]) & (i < r[2]) & (j >= r[3]) & (j < r[4]) & (k >= r[5]) & (k < r[6])
                mat = r[7:]
            else:
                sel = k + 0.5 < r[1] + r[2] * (i + 0.5) + r[3] * (j + 0.5)
                mat = r[4:]
            for a, v in zip(out, mat):
                a[sel] = v
        if grade is not None:
            out[0] = (out[0].astype(np.float64) * grade).astype(np.float32)
            out[1] = (out[1].astype(np.float64) * grade).astype(np.float32)
            out[2] = (out[2].astype(np.float64) * (1.0 + (grade - 1.0) / 2.0)).astype(np.float32)
    return out[0], out[1], out[2], 1000.0 / n


def np8_stripe(g, step, rank, n):
    """(tm2, tm1) of one rank from its raw checkpoint stripe (io_checkpoint.c:93-118)."""
    s = g["ckpt%d_stripe_%d" % (int(step), rank)]
    return s[:3 * n].reshape(n, 3), s[3 * n:6 * n].reshape(n, 3)


# ---------------------------------------------------------------------------------------------
# dependency-cone windows of an octree mesh (oracle parity at sizes the oracle cannot run whole): oracle/windows.py
# ---------------------------------------------------------------------------------------------
from oracle.windows import hanging_kinds, lateral_windows, octree_window, octree_window_oracle    # noqa: E402,F401


def lap_timer(name):
    """-> lap(what): with HQ_TEST_LAPS=1 prints where a long test's time goes (stderr; run pytest with -s)."""
    import os
    import sys
    import time
    t = [time.time()]

    def lap(what):
        if os.environ.get("HQ_TEST_LAPS"):
            now = time.time()
            sys.stderr.write("  [%s] %-40s %7.1f s\n" % (name, what, now - t[0]))
            t[0] = now
    return lap


def two_material_leaves(nx=128, ny=32, nz=32, split=37, h=100.0):
    """Leaves (pre-order) of a uniform nx x ny x nz box whose material changes at the element column i = split -- a material
    boundary INSIDE a level that is not aligned with the 64-wide brick tiles: the tile that straddles it holds simple nodes
    of two materials.  -> (elem_ticks, elem_edge, edata [E,4] = h, Vp, Vs, rho, far_ticks)"""
    e = 1 << 23
    i, j, k = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    i, j, k = i.ravel(), j.ravel(), k.ravel()
    order = np.argsort(ho.zvalue(i, j, k), kind="stable")
    i, j, k = i[order], j[order], k[order]
    ticks = (np.stack([i, j, k], 1).astype(np.int64) * e).astype(np.uint32)
    edata = np.empty((len(i), 4), np.float32)
    edata[:] = (h, 6000.0, 3464.0, 2700.0)
    edata[i < split] = (h, 3000.0, 1732.0, 2200.0)
    return ticks, np.full(len(i), e, np.uint32), edata, (nx * e, ny * e, nz * e)
