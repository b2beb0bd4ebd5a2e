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


def c1_problem(damping="rayleigh"):
    lnid, node_ijk, elem_ijk, edata = c1_mesh()
    face = ho.face_bits(elem_ijk, C1_NX, C1_NY, C1_NZ)
    etable, ntable = ho.solver_init(lnid, edata.copy(), face, len(node_ijk), 1e-3, 5.0,
                                    damping=ho.DAMPING_BY_NAME[damping])
    return dict(lnid=lnid, node_ijk=node_ijk, elem_ijk=elem_ijk, edata=edata, face=face,
                etable=etable, ntable=ntable, N=len(node_ijk), E=len(lnid), dt=1e-3,
                damping=ho.DAMPING_BY_NAME[damping])


def rel_linf(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300))


def c5_problem():
    """The reference's two-level mesh (soft layer refined one level deeper, 800 hanging
    nodes) rebuilt from its flat dump, with eTable / nTable as solver_init leaves them
    (incl. the hanging-node mass distribution, psolve.c:3498-3507)."""
    g = load("c5_two_level")
    m = ho.octree_mesh_from_elem_ticks(g["elem_ticks"], C1_FAR_TICKS)
    E, N = len(m["lnid"]), len(m["node_q"])
    mat = g["mat_vs_vp_rho"]
    tick = 1000.0 / 2 ** 30
    edata = np.empty((E, 4), np.float32)
    edata[:, 0] = (tick * m["emin"] * m["elem_size"].astype(np.float64)).astype(np.float32)
    edata[:, 1], edata[:, 2], edata[:, 3] = mat[:, 1], mat[:, 0], mat[:, 2]
    etable, ntable = ho.solver_init(m["lnid"], edata, m["face"], N, 1e-3, 5.0)
    ho.compute_adjust(ntable, 0, m["dangling"])
    return dict(lnid=m["lnid"], node_q=m["node_q"], etable=etable, ntable=ntable, dangling=m["dangling"],
                N=N, E=E, dt=1e-3, emin=m["emin"], golden=g)
