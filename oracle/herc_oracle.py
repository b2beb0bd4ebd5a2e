"""ctypes front-end of oracle/herc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  hercules_amd/ never does.
"""
import ctypes
import os
import subprocess

import math
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "herc_oracle.c")
_LIB = os.path.join(_HERE, "libherc_oracle.so")

DAMP_NONE, DAMP_RAYLEIGH, DAMP_MASS = 0, 1, 2
STIFF_EFFECTIVE, STIFF_CONVENTIONAL = 0, 1
DAMPING_BY_NAME = {"none": DAMP_NONE, "rayleigh": DAMP_RAYLEIGH, "mass": DAMP_MASS}

_LIB_F32 = os.path.join(_HERE, "libherc_oracle_f32.so")      # the same file with -DSINGLE_PRECISION_SOLVER (psolve.h:60-64)

_lib = None
_lib_f32 = None


def build(force=False):
    """Compile the C restatement with gcc (seconds): solver_float = double, and = float as the reference's
    -DSINGLE_PRECISION_SOLVER build has it."""
    for out, defs in ((_LIB, []), (_LIB_F32, ["-DSINGLE_PRECISION_SOLVER"])):
        if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(_SRC):
            subprocess.check_call(["gcc", "-O2", "-std=c99", "-fPIC", "-shared", "-fvisibility=hidden",
                                   "-fopenmp"] + defs + ["-o", out, _SRC, "-lm"])
    return _LIB


def _open(path):
    L = ctypes.CDLL(path)
    L.ho_uniform_mesh.restype = ctypes.c_int
    L.ho_solver_init.restype = ctypes.c_int64
    L.ho_zvalue.restype = ctypes.c_uint64
    L.ho_real_bytes.restype = ctypes.c_int32
    return L


def lib(real=np.float64):
    """The library whose solver_float is `real` (float64: the default build; float32: -DSINGLE_PRECISION_SOLVER)."""
    global _lib, _lib_f32
    if np.dtype(real) == np.float32:
        if _lib_f32 is None:
            build()
            _lib_f32 = _open(_LIB_F32)
            assert _lib_f32.ho_real_bytes() == 4
        return _lib_f32
    assert np.dtype(real) == np.float64
    if _lib is None:
        build()
        _lib = _open(_LIB)
        assert _lib.ho_real_bytes() == 8
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _c(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    return a


def uniform_mesh(nx, ny, nz):
    E = nx * ny * nz
    N = (nx + 1) * (ny + 1) * (nz + 1)
    elem_ijk = np.empty((E, 3), np.int32)
    lnid = np.empty((E, 8), np.int32)
    node_ijk = np.empty((N, 3), np.int32)
    rc = lib().ho_uniform_mesh(ctypes.c_int32(nx), ctypes.c_int32(ny), ctypes.c_int32(nz),
                               _p(elem_ijk), _p(lnid), _p(node_ijk))
    if rc != 0:
        raise MemoryError("ho_uniform_mesh")
    return elem_ijk, lnid, node_ijk


def face_bits(elem_ijk, nx, ny, nz):
    """Six 'touches the domain face' bits per element (compute_setflag inputs,
    psolve.c:3414-3426): bit0-2 near x,y,z; bit3-5 far x,y,z."""
    e = np.asarray(elem_ijk)
    f = (e[:, 0] == 0) * 1 | (e[:, 1] == 0) * 2 | (e[:, 2] == 0) * 4 | \
        (e[:, 0] == nx - 1) * 8 | (e[:, 1] == ny - 1) * 16 | (e[:, 2] == nz - 1) * 32
    return f.astype(np.uint8)


def compute_K():
    K1 = np.zeros((8, 8, 3, 3))
    K2 = np.zeros((8, 8, 3, 3))
    lib().ho_compute_K(_p(K1), _p(K2))
    return K1, K2


def setab(freq, damping):
    a = ctypes.c_double()
    b = ctypes.c_double()
    lib().ho_setab(ctypes.c_double(freq), ctypes.c_int(damping), ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def solver_init(lnid, edata, face, N, dt, freq, damping=DAMP_RAYLEIGH, thr_damping=0.05,
                thr_vpvs=3.0, boundary=True, halfspace=True, real=np.float64):
    """-> etable [E,4], ntable [N,7]; edata [E,4] float32 (h,Vp,Vs,rho), may be modified.
    real: solver_float -- the n_t rows are accumulated in it (float32: as psolve_f32 does)."""
    lnid = _c(lnid, np.int32)
    E = lnid.shape[0]
    assert edata.dtype == np.float32 and edata.flags.c_contiguous
    face = _c(face, np.uint8)
    etable = np.zeros((E, 4))
    ntable = np.zeros((N, 7), real)
    rc = lib(real).ho_solver_init(ctypes.c_int64(E), ctypes.c_int64(N), _p(lnid), _p(edata), _p(face),
                              ctypes.c_double(dt), ctypes.c_double(freq), ctypes.c_int(damping),
                              ctypes.c_double(thr_damping), ctypes.c_double(thr_vpvs),
                              ctypes.c_int(int(boundary)), ctypes.c_int(int(halfspace)),
                              _p(etable), _p(ntable))
    if rc != 0:
        raise ValueError("element %d produces negative lambda" % (rc - 1))
    return etable, ntable


def solver_run(lnid, etable, ntable, tm1, tm2, step0, nsteps, dt, damping=DAMP_RAYLEIGH,
               stiff_method=STIFF_EFFECTIVE, formulation=0, zero_skip=True, loaded_lnid=None,
               forces=None, cap_lnid=None, K=None, dangling=None):
    """March in place; returns captured tm1 rows [nsteps, cap_n, 3] (or None)."""
    lnid = _c(lnid, np.int32)
    E = lnid.shape[0]
    N = ntable.shape[0]
    real = tm1.dtype                 # solver_float: float64, or float32 = the reference's -DSINGLE_PRECISION_SOLVER
    assert etable.dtype == np.float64 and etable.flags.c_contiguous
    for a in (ntable, tm1, tm2):
        assert a.dtype == real and a.flags.c_contiguous
    K1, K2 = K if K is not None else compute_K()
    force = np.zeros((N, 3), real)
    if loaded_lnid is None or len(loaded_lnid) == 0:
        nl, ll, F, nfs = 0, None, None, 0
    else:
        ll = _c(loaded_lnid, np.int32)
        F = _c(forces, np.float64)
        nl, nfs = len(ll), F.shape[0]
    if cap_lnid is None or len(cap_lnid) == 0:
        cn, cl, cap = 0, None, None
    else:
        cl = _c(np.asarray(cap_lnid).reshape(-1), np.int32)
        cn = len(cl)
        cap = np.zeros((nsteps, cn, 3))
    if dangling is None:
        nd, di, dp, da = 0, None, None, None
    else:
        di, dp, da = [_c(x, np.int32) for x in dangling]
        nd = len(di)
    lib(real).ho_solver_run(ctypes.c_int64(E), ctypes.c_int64(N), _p(lnid), _p(etable), _p(ntable),
                        _p(K1), _p(K2), _p(tm1), _p(tm2), _p(force), ctypes.c_int32(step0),
                        ctypes.c_int32(nsteps), ctypes.c_double(dt), ctypes.c_int(damping),
                        ctypes.c_int(stiff_method), ctypes.c_int(formulation),
                        ctypes.c_int(int(zero_skip)), ctypes.c_int32(nl), _p(ll), _p(F),
                        ctypes.c_int32(nfs), ctypes.c_int32(cn), _p(cl), _p(cap),
                        ctypes.c_int32(nd), _p(di), _p(dp), _p(da))
    return cap


def compute_adjust(table, how, dangling):
    """compute_adjust (psolve.c:5936-6039) in place; how 0 = DISTRIBUTION, 1 = ASSIGNMENT."""
    di, dp, da = [_c(x, np.int32) for x in dangling]
    assert table.dtype in (np.float64, np.float32) and table.flags.c_contiguous
    lib(table.dtype).ho_compute_adjust(_p(table), ctypes.c_int32(table.shape[1]), ctypes.c_int32(how),
                            ctypes.c_int32(len(di)), _p(di), _p(dp), _p(da))


def octree_mesh_from_elem_ticks(elem_ticks, far_ticks):
    """Rebuild octor's tables from the reference's flat element dump for meshes with
    several refinement levels (no buildings):
      nodes   : unique vertices in Z-order of the far-boundary-adjusted coordinates
                (octor.c:6100-6106, 6166)
      dangling: node_setproperty (octor.c:3280-3860) by touch count / boundary position /
                alignment to the next coarser grid; anchors as dnode correlation lists them
                (octor.c:6493-6612, prepended => reversed order)
    -> dict(lnid, node_q, elem_q, elem_size, emin, face, dangling=(ids, ptr, anchors))
    where *_q are coordinates in units of the smallest edge."""
    t = np.asarray(elem_ticks, np.int64)
    edge = t[:, 1, 0] - t[:, 0, 0]
    emin = int(edge.min())
    assert np.all(t % emin == 0)
    q = t // emin
    size = (edge // emin).astype(np.int64)
    nq = [int(f // emin) for f in far_ticks]
    flat = q.reshape(-1, 3)
    uniq, inv = np.unique(flat, axis=0, return_inverse=True)
    key_c = [np.where(uniq[:, d] == nq[d], 2 * nq[d] - 1, 2 * uniq[:, d]) for d in range(3)]
    order = np.argsort(zvalue(key_c[0], key_c[1], key_c[2]), kind="stable")
    rank = np.empty(len(uniq), np.int64)
    rank[order] = np.arange(len(uniq))
    lnid = rank[inv.reshape(-1)].reshape(-1, 8).astype(np.int32)
    node_q = uniq[order]
    N = len(node_q)
    ll = q[:, 0, :]
    face = ((ll[:, 0] == 0) * 1 | (ll[:, 1] == 0) * 2 | (ll[:, 2] == 0) * 4 |
            (ll[:, 0] + size == nq[0]) * 8 | (ll[:, 1] + size == nq[1]) * 16 |
            (ll[:, 2] + size == nq[2]) * 32).astype(np.uint8)
    touches = np.bincount(lnid.reshape(-1), minlength=N)
    small = np.full(N, np.iinfo(np.int64).max)
    np.minimum.at(small, lnid.reshape(-1), np.repeat(size, 8))     # vertex->level: smallest toucher
    where = sum(((node_q[:, d] == 0) | (node_q[:, d] == nq[d])).astype(int) for d in range(3))
    index = {tuple(v): i for i, v in enumerate(node_q.tolist())}
    ids, ptr, anchors = [], [0], []
    for n in range(N):
        tc, wh, s = int(touches[n]), int(where[n]), int(small[n])
        if tc == 8 or (tc == 4 and wh == 1) or (tc == 2 and wh == 2) or (tc == 1 and wh == 3):
            continue                                             # anchored
        mods = [int(node_q[n, d] % (2 * s) != 0) for d in range(3)]
        nm = sum(mods)
        ok = (tc == 6 and wh == 0 and nm == 1) or (tc == 4 and wh == 0 and nm in (1, 2)) or \
             (tc == 2 and wh in (0, 1) and nm == 1)
        if not ok:
            raise ValueError("node %d: touches %d where %d mods %s is not a legal octor vertex" % (n, tc, wh, mods))
        pts = []
        if nm == 1:                                              # X/Y/ZEDGE: -s then +s
            d = mods.index(1)
            for sg in (-s, s):
                p = node_q[n].copy(); p[d] += sg; pts.append(tuple(p))
        else:                                                    # face: the two in-plane axes, low axis fastest
            a, b = [d for d in range(3) if mods[d]]
            for dep in range(4):
                p = node_q[n].copy()
                p[a] += s if (dep & 1) else -s
                p[b] += s if (dep & 2) else -s
                pts.append(tuple(p))
        ids.append(n)
        anchors += [index[p] for p in reversed(pts)]             # the list is built by prepending
        ptr.append(len(anchors))
    return dict(lnid=lnid, node_q=node_q.astype(np.int32), elem_q=ll.astype(np.int32),
                elem_size=size.astype(np.int32), emin=emin, face=face,
                dangling=(np.array(ids, np.int32), np.array(ptr, np.int32), np.array(anchors, np.int32)))


# ---------------------------------------------------------------------------
# numpy glue shared by the tests
# ---------------------------------------------------------------------------

def zvalue(x, y, z):
    """Vectorised ho_zvalue for small non-negative ints (< 2**21)."""
    def spread(v):
        v = np.asarray(v, np.uint64) & np.uint64(0x1fffff)
        v = (v | (v << np.uint64(32))) & np.uint64(0x1f00000000ffff)
        v = (v | (v << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
        v = (v | (v << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
        v = (v | (v << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
        return v
    return spread(x) | (spread(y) << np.uint64(1)) | (spread(z) << np.uint64(2))


def mesh_from_elem_ticks(elem_ticks, far_ticks):
    """Rebuild octor's node numbering from the reference's flat element dump
    (meshformatlab.c:52-250; uniform meshes): unique nodes sorted by the Z-value
    of their far-boundary-adjusted coordinates (octor.c:6100-6106, 6166).
    -> lnid [E,8], node_ijk [N,3] (element units), elem_ijk [E,3], edge_ticks"""
    t = np.asarray(elem_ticks, np.int64)
    edge = int(t[0, 1, 0] - t[0, 0, 0])
    assert np.all(t[:, 1, 0] - t[:, 0, 0] == edge), "uniform meshes only"
    ijk = t // edge
    n_ax = [int(f // edge) for f in far_ticks]
    flat = ijk.reshape(-1, 3)
    uniq, inv = np.unique(flat, axis=0, return_inverse=True)
    key_c = [np.where(uniq[:, d] == n_ax[d], 2 * n_ax[d] - 1, 2 * uniq[:, d]) for d in range(3)]
    order = np.argsort(zvalue(key_c[0], key_c[1], key_c[2]), kind="stable")
    rank = np.empty(len(uniq), np.int64)
    rank[order] = np.arange(len(uniq))
    lnid = rank[inv.reshape(-1)].reshape(-1, 8).astype(np.int32)
    return lnid, uniq[order].astype(np.int32), ijk[:, 0, :].astype(np.int32), edge


def station_weights(points_m, h, nx, ny, nz, lnid, elem_ijk):
    """Containing element + trilinear weights for output stations
    (compute_csi_eta_dzeta psolve.c:6378-6440; interpolate_station_displacements
    :6679-6710).  -> node ids [S,8], phi [S,8]"""
    lut = {tuple(v): i for i, v in enumerate(np.asarray(elem_ijk).tolist())}
    ids, phis = [], []
    for p in np.asarray(points_m, float):
        e_ijk = [min(int(np.floor(p[d] / h)), n - 1) for d, n in enumerate((nx, ny, nz))]
        e = lut[tuple(e_ijk)]
        loc = [2 * (p[d] - h * (e_ijk[d] + 0.5)) / h for d in range(3)]
        phi = [(1 + (1 if (n >> 0) & 1 else -1) * loc[0]) * (1 + (1 if (n >> 1) & 1 else -1) * loc[1])
               * (1 + (1 if (n >> 2) & 1 else -1) * loc[2]) / 8 for n in range(8)]
        ids.append(lnid[e])
        phis.append(phi)
    return np.array(ids, np.int32), np.array(phis)


def station_kinematics(phi, tm1, tm2=None, tm3=None, dt=1.0, derivs=0):
    """interpolate_station_displacements (psolve.c:6705-6787) for one station: phi [8], tm* [8,3]
    rows of its element's nodes.  Displacement = sum over the nodes; the velocity takes phi * tm2 off
    the same accumulator node by node and divides by dt; the acceleration takes phi * tm2 off once
    more, adds phi * tm3, over dt^2.  -> 3 (1 + derivs) values."""
    d = [0.0, 0.0, 0.0]
    for c in range(8):
        for a in range(3):
            d[a] += float(phi[c]) * float(tm1[c][a])
    out = list(d)
    if derivs >= 1:
        for c in range(8):
            for a in range(3):
                d[a] -= float(phi[c]) * float(tm2[c][a])
        out += [d[a] / dt for a in range(3)]
    if derivs == 2:
        for c in range(8):
            for a in range(3):
                d[a] -= float(phi[c]) * float(tm2[c][a])
                d[a] += float(phi[c]) * float(tm3[c][a])
        out += [d[a] / (dt * dt) for a in range(3)]
    return np.array(out)


def station_line(time, vals):
    """A data line of a station file (psolve.c:6727-6787): newline first, "%10.6f", then "% 8e" each."""
    return "\n%10.6f" % time + "".join(" % 8e" % v for v in vals)


def station_header(derivs=0):
    """psolve.c:6636-6648."""
    return ("#  Time(s)         X|(m)         Y-(m)         Z.(m)"
            + ("       X|(m/s)       Y-(m/s)       Z.(m/s)" if derivs >= 1 else "")
            + ("      X|(m/s2)      Y-(m/s2)      Z.(m/s2)" if derivs == 2 else ""))


def domain_coords_linearinterp(lon, lat, lon_corners, lat_corners, len_eta, len_csi):
    """(longitude, latitude) -> domain x, y: Newton iteration on the bilinear map of the four
    surface corners (compute_domain_coords_linearinterp, geometrics.c:178-244)."""
    X, Y = lat, lon
    Xi, Yi = [float(v) for v in lat_corners], [float(v) for v in lon_corners]
    Ax = 4 * X - (Xi[0] + Xi[1] + Xi[2] + Xi[3]); Ay = 4 * Y - (Yi[0] + Yi[1] + Yi[2] + Yi[3])
    Bx = -Xi[0] + Xi[1] + Xi[2] - Xi[3];          By = -Yi[0] + Yi[1] + Yi[2] - Yi[3]
    Cx = -Xi[0] - Xi[1] + Xi[2] + Xi[3];          Cy = -Yi[0] - Yi[1] + Yi[2] + Yi[3]
    Dx = Xi[0] - Xi[1] + Xi[2] - Xi[3];           Dy = Yi[0] - Yi[1] + Yi[2] - Yi[3]
    xn = [0.0, 0.0]
    res = 1e10
    while res > 1e-6:
        m00 = Bx + Dx * xn[1]; m01 = Cx + Dx * xn[0]
        m10 = By + Dy * xn[1]; m11 = Cy + Dy * xn[0]
        f0 = -Ax + Bx * xn[0] + Cx * xn[1] + Dx * xn[0] * xn[1]
        f1 = -Ay + By * xn[0] + Cy * xn[1] + Dy * xn[0] * xn[1]
        det = m00 * m11 - m10 * m01
        d0 = -(f0 * m11 - f1 * m01) / det
        d1 = -(f1 * m00 - f0 * m10) / det
        res = abs(f0) + abs(f1)
        xn[0] += d0; xn[1] += d1
    return 0.5 * (xn[0] + 1) * len_csi, 0.5 * (xn[1] + 1) * len_eta


def plane_points(origin_xyz, step_strike, n_strike, step_dip, n_dip, strike_deg, dip_deg):
    """Grid of an output plane in domain coordinates, index = iStrike * n_dip + iDownDip
    (Old_output_planes_construct_strips io_planes.c:489-520; compute_global_coords with
    rake = 0, geometrics.c:33-70).  -> [n_strike * n_dip, 3]"""
    PI = 3.14159265358979323846
    d, l, p = dip_deg * PI / 180, 0.0 * PI / 180, strike_deg * PI / 180
    out = np.empty((n_strike * n_dip, 3))
    for i in range(n_strike):
        for j in range(n_dip):
            x, y, z = i * step_strike, j * step_dip, 0.0
            gx = (math.cos(p) * math.cos(l) + math.sin(p) * math.cos(d) * math.sin(l)) * x \
                - (-math.cos(p) * math.sin(l) + math.sin(p) * math.cos(d) * math.cos(l)) * y \
                - (-math.sin(p) * math.sin(d)) * z
            gy = (math.sin(p) * math.cos(l) - math.cos(p) * math.cos(d) * math.sin(l)) * x \
                - (-math.sin(p) * math.sin(l) - math.cos(p) * math.cos(d) * math.cos(l)) * y \
                - (math.cos(p) * math.sin(d)) * z
            gz = -math.sin(d) * math.sin(l) * x + math.sin(d) * math.cos(l) * y + math.cos(d) * z
            out[i * n_dip + j] = (gx + origin_xyz[0], gy + origin_xyz[1], gz + origin_xyz[2])
    return out


def plane_displacements(tm1, ids, phi):
    """Old_planes_print (io_planes.c:151-200): sum over the 8 nodes in lnid order, per component."""
    out = np.zeros((len(ids), 3))
    for c in range(8):
        out += phi[:, c:c + 1] * tm1[ids[:, c]]
    return out


# ---------------------------------------------------------------------------
# octor's multi-rank tables for an octree mesh, from the global view
# ---------------------------------------------------------------------------

def octree_partition(m, nranks, far_q):
    """Restates what octor_extractmesh leaves on every rank for the mesh `m`
    (octree_mesh_from_elem_ticks) cut into `nranks` blocks:

      elements : contiguous blocks of the pre-ordered leaves (BLOCK_LOW/HIGH, octor.c:4939-4944)
      owner    : rank of the leaf that contains the far-boundary-adjusted node (octor.c:5466-5475)
      harbored : vertices of the rank's elements + the nodes it owns (direct sharing,
                 octor.c:5516-5793) + the anchors of the hanging nodes it owns (indirect
                 sharing, node_harboranchored octor.c:3916-4042, :5795-6040); Z-ordered
      sharers  : ranks that have an owned node as element vertex, or harbor it as an anchor
      dnodeTable: the hanging nodes the rank OWNS, anchors as local ids
      an_sched / dn_sched: schedule_build (psolve.c:4704-4863); messengers in the reference's list order

    -> list of per-rank dicts."""
    lnid, node_q, elem_q, size = m["lnid"], m["node_q"], m["elem_q"], m["elem_size"]
    E, N = len(lnid), len(node_q)
    dn_ids, dn_ptr, dn_anc = m["dangling"]
    dn_of = np.full(N, -1)
    dn_of[dn_ids] = np.arange(len(dn_ids))
    erank = ((np.arange(E, dtype=np.int64) + 1) * nranks - 1) // E
    # leaf containing each fine cell
    cell = np.full((far_q[0], far_q[1], far_q[2]), -1, np.int64)
    for e in range(E):
        i, j, k = (int(v) for v in elem_q[e])
        s = int(size[e])
        cell[i:i + s, j:j + s, k:k + s] = e
    adj = np.minimum(node_q, np.array(far_q) - 1)
    owner = erank[cell[adj[:, 0], adj[:, 1], adj[:, 2]]]
    verts = [set(lnid[erank == r].reshape(-1).tolist()) for r in range(nranks)]
    out = []
    harbored = []
    for r in range(nranks):
        h = set(verts[r]) | set(np.nonzero(owner == r)[0].tolist())
        for k in np.nonzero(owner[dn_ids] == r)[0]:
            h |= set(dn_anc[dn_ptr[k]:dn_ptr[k + 1]].tolist())
        harbored.append(h)
    for r in range(nranks):
        nodes = np.array(sorted(harbored[r]), np.int64)              # global ids are already Z-ordered
        loc = {int(g): i for i, g in enumerate(nodes)}
        e_ids = np.nonzero(erank == r)[0]
        l_lnid = np.array([[loc[int(g)] for g in lnid[e]] for e in e_ids], np.int32).reshape(-1, 8)
        own = owner[nodes]
        # hanging nodes owned here, in local order
        d_loc, d_ptr, d_anc = [], [0], []
        for i, g in enumerate(nodes):
            k = dn_of[g]
            if k >= 0 and own[i] == r:
                d_loc.append(i)
                d_anc += [loc[int(a)] for a in dn_anc[dn_ptr[k]:dn_ptr[k + 1]]]
                d_ptr.append(len(d_anc))
        # the order in which this rank met its neighbours (com_allocpctl, octor.c:2640-2741): its leaves in order, around each
        # the 4 x 4 x 4 points at half-edge spacing from (corner - edge / 2), z outermost; the rank that holds a point's pixel
        met = []
        for e in e_ids:
            sz = float(size[e])
            for kk in range(4):
                z = elem_q[e][2] - sz / 2 + sz / 2 * kk
                if z < 0 or z >= far_q[2]:
                    continue
                for jj in range(4):
                    y = elem_q[e][1] - sz / 2 + sz / 2 * jj
                    if y < 0 or y >= far_q[1]:
                        continue
                    for ii in range(4):
                        x = elem_q[e][0] - sz / 2 + sz / 2 * ii
                        if x < 0 or x >= far_q[0]:
                            continue
                        q = int(erank[cell[int(math.floor(x)), int(math.floor(y)), int(math.floor(z))]])
                        if q != r and q not in met:
                            met.append(q)
        # schedules.  A vertex's share list (octor.c:5700-5793, 5990-6050): the ranks that have it as an element vertex, in the
        # order this rank met them (their messages are taken in the reverse order and each sender goes to the HEAD of the
        # list); ahead of them the ranks that hold it only as an anchor of a hanging node of theirs, in descending rank
        sched = {"an": {"c": {}, "s": {}}, "dn": {"c": {}, "s": {}}}
        for i, g in enumerate(nodes):
            kind = "dn" if dn_of[g] >= 0 else "an"
            if own[i] != r:
                sched[kind]["c"].setdefault(int(own[i]), []).append(i)
            else:
                direct = [q for q in met if int(g) in verts[q]]
                indirect = [q for q in range(nranks - 1, -1, -1) if q != r and int(g) in harbored[q] and int(g) not in verts[q]]
                for q in indirect + direct:
                    sched[kind]["s"].setdefault(q, []).append(i)
                assert len(indirect) + len(direct) == sum(1 for q in range(nranks) if q != r and int(g) in harbored[q])
        # schedule_build (psolve.c:4711-4795) walks the nodes in local order and a node's share list (above), and
        # puts a NEW messenger at the HEAD of its list: the lists end up in the reverse of the order of first encounter.
        # schedule_senddata adds the incoming records messenger by messenger in list order (:5035-5073) -- with that order
        # the multi-rank runs below are bit-identical to the reference's per-rank checkpoint stripes, float and double
        pack = lambda d: [(q, np.array(v, np.int32)) for q, v in reversed(list(d.items()))]
        out.append(dict(rank=r, elems=e_ids, nodes=nodes, owner=own.astype(np.int32), lnid=l_lnid,
                        dangling=(np.array(d_loc, np.int32), np.array(d_ptr, np.int32), np.array(d_anc, np.int32)),
                        an_sched={k: pack(v) for k, v in sched["an"].items()},
                        dn_sched={k: pack(v) for k, v in sched["dn"].items()},
                        is_dangling=(dn_of[nodes] >= 0)))
    return out


def _exchange_sim(parts, tables, kind, contribution):
    """schedule_senddata (psolve.c:4945-5079) between in-memory ranks."""
    msgs = {}
    for p in parts:
        for q, mapping in p[kind]["c" if contribution else "s"]:
            msgs[(p["rank"], q)] = tables[p["rank"]][mapping].copy()
    for p in parts:
        for q, mapping in p[kind]["s" if contribution else "c"]:
            rec = msgs[(q, p["rank"])]
            if contribution:
                np.add.at(tables[p["rank"]], mapping, rec)
            else:
                tables[p["rank"]][mapping] = rec


def multi_rank_init(parts, edata_by_rank, face_by_rank, dt, freq, **kw):
    """solver_init on every rank incl. the three-stage mass exchange (psolve.c:3498-3507)."""
    ets, nts = [], []
    for p, ed, fc in zip(parts, edata_by_rank, face_by_rank):
        et, nt = solver_init(p["lnid"], ed, fc, len(p["nodes"]), dt, freq, **kw)
        ets.append(et)
        nts.append(nt)
    _exchange_sim(parts, nts, "dn_sched", True)
    for p, nt in zip(parts, nts):
        if len(p["dangling"][0]):
            compute_adjust(nt, 0, p["dangling"])
    _exchange_sim(parts, nts, "an_sched", True)
    return ets, nts


def multi_rank_run(parts, ets, nts, tm1s, tm2s, step0, nsteps, dt, loaded, forces):
    """solver_run (psolve.c:4265-4319) for all ranks in lockstep; tm1s/tm2s are the
    reference's pre-swap arrays per rank, updated in place."""
    real = tm1s[0].dtype                  # solver_float: the records of every exchange have it (psolve.c:4985-5073)
    L = lib(real)
    K1, K2 = compute_K()
    c64 = ctypes.c_int64
    frc = [np.zeros((len(p["nodes"]), 3), real) for p in parts]
    a = [t for t in tm1s]
    b = [t for t in tm2s]
    for step in range(step0, step0 + nsteps):
        a, b = b, a                                            # psolve.c:4271-4273
        for p, et, u1, u2, f, ld, F in zip(parts, ets, a, b, frc, loaded, forces):
            if len(ld) and step < len(F):
                L.ho_addforce_source(ctypes.c_int32(len(ld)), _p(_c(ld, np.int32)), _p(np.ascontiguousarray(F[step])),
                                     ctypes.c_double(dt * dt), _p(f))
            n_e = len(p["lnid"])
            L.ho_addforce_effective(c64(n_e), _p(p["lnid"]), _p(et), _p(u1), _p(f), 1)
            L.ho_damping_addforce(c64(n_e), _p(p["lnid"]), _p(et), _p(u1), _p(u2), _p(K1), _p(K2), _p(f), 1)
        _exchange_sim(parts, frc, "dn_sched", True)             # :4298
        for p, f in zip(parts, frc):                            # :4299
            if len(p["dangling"][0]):
                compute_adjust(f, 0, p["dangling"])
        _exchange_sim(parts, frc, "an_sched", True)             # :4301
        for p, nt, u1, u2, f in zip(parts, nts, a, b, frc):     # :4305
            L.ho_compute_displacement(c64(len(p["nodes"])), _p(nt), _p(u1), _p(u2), _p(f), None)
        _exchange_sim(parts, b, "an_sched", False)              # :4312
        for p, u2 in zip(parts, b):                             # :4313
            if len(p["dangling"][0]):
                compute_adjust(u2, 1, p["dangling"])
        _exchange_sim(parts, b, "dn_sched", False)              # :4315
    if nsteps % 2:                                              # keep the callers' arrays in their roles
        for t1, t2 in zip(tm1s, tm2s):
            tmp = t1.copy()
            t1[:] = t2
            t2[:] = tmp
