"""ctypes front-end of oracle/herc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  hercules_amd/ never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "herc_oracle.c")
_LIB = os.path.join(_HERE, "libherc_oracle.so")

DAMP_NONE, DAMP_RAYLEIGH, DAMP_MASS = 0, 1, 2
STIFF_EFFECTIVE, STIFF_CONVENTIONAL = 0, 1
DAMPING_BY_NAME = {"none": DAMP_NONE, "rayleigh": DAMP_RAYLEIGH, "mass": DAMP_MASS}

_lib = None


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O2", "-std=c99", "-fPIC", "-shared", "-fvisibility=hidden",
                               "-fopenmp", "-o", _LIB, _SRC, "-lm"])
    return _LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB)
        _lib.ho_uniform_mesh.restype = ctypes.c_int
        _lib.ho_solver_init.restype = ctypes.c_int64
        _lib.ho_zvalue.restype = ctypes.c_uint64
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
                thr_vpvs=3.0, boundary=True, halfspace=True):
    """-> etable [E,4], ntable [N,7]; edata [E,4] float32 (h,Vp,Vs,rho), may be modified."""
    lnid = _c(lnid, np.int32)
    E = lnid.shape[0]
    assert edata.dtype == np.float32 and edata.flags.c_contiguous
    face = _c(face, np.uint8)
    etable = np.zeros((E, 4))
    ntable = np.zeros((N, 7))
    rc = lib().ho_solver_init(ctypes.c_int64(E), ctypes.c_int64(N), _p(lnid), _p(edata), _p(face),
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
    for a in (etable, ntable, tm1, tm2):
        assert a.dtype == np.float64 and a.flags.c_contiguous
    K1, K2 = K if K is not None else compute_K()
    force = np.zeros((N, 3))
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
    lib().ho_solver_run(ctypes.c_int64(E), ctypes.c_int64(N), _p(lnid), _p(etable), _p(ntable),
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
    assert table.dtype == np.float64 and table.flags.c_contiguous
    lib().ho_compute_adjust(_p(table), ctypes.c_int32(table.shape[1]), ctypes.c_int32(how),
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
