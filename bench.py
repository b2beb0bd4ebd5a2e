#!/usr/bin/env python3
"""bench.py -- element-updates/s of the explicit time-stepping hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c1]

One "step" = one iteration of the solver_run loop body (psolve.c:4265-4319) over
the whole mesh: source force, stiffness + Rayleigh damping element force, nodal
central-difference update (and the halo exchange when N > 1).

Workload (BASELINE.json configs[2], the one the metric is quoted on): 64 M-element
uniform box 512 x 512 x 256, homogeneous half-space Vp 6000 / Vs 3464 / rho 2700,
Rayleigh damping, Lysmer dashpots on five faces, point double-couple source,
started from a seeded random displacement field so every element is active.
N > 1 partitions the SAME mesh along the octor block decomposition (strong
scaling) and exchanges interface-node forces / displacements with RCCL.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# HBM bytes per element-update of the REFERENCE's array-at-a-time formulation (fp64,
# uniform mesh).  The fused patch kernel never moves the force vector, so this figure is NOT a bound
# on it: it is reported only as `algorithmic_equiv_GBs`, never as roofline.frac.
BYTES_PER_ELEMENT_UPDATE = 336.0
BYTES_ELEMENT_PHASE = 160.0          # lnid 32 + coefficients 32 + tm1,tm2 48 + force RMW 48
# what ANY formulation must move per node and step: read u(t), u(t-dt), write u(t+dt)
COMPULSORY_BYTES_PER_NODE = 72.0
# a mesh whose material differs from element to element also has to read every node's own n_t row (24 B in the
# 3-double form) and every element's (c1, c2, beta) (24 B; elements ~ nodes): 120 B per node and step
COMPULSORY_BYTES_PER_NODE_LATERAL = 120.0
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8 TB/s

WORKLOADS = {
    # name: (nx, ny, nz, h, dt, freq)
    "c3": (512, 512, 256, 1000.0 / 512, 9.0e-5, 200.0),
    "c2": (256, 256, 128, 1000.0 / 256, 1.8e-4, 100.0),
    "c1": (16, 16, 8, 62.5, 1.0e-3, 5.0),
    # c3 / c2 / m1 with material that differs from element to element (LATERAL): what solver_init hands over on any
    # real CVM mesh (psolve.c:3360-3409), and what no uniform-coefficient fast path applies to
    "c3h": (512, 512, 256, 1000.0 / 512, 9.0e-5, 200.0),
    "c2h": (256, 256, 128, 1000.0 / 256, 1.8e-4, 100.0),
    "m1h": (128, 128, 64, 1000.0 / 128, 3.6e-4, 50.0),
    "m1": (128, 128, 64, 1000.0 / 128, 3.6e-4, 50.0),
    # two-level octree box (hanging nodes): nx, ny, nz_fine, h_fine, dt, freq  (+ 96 coarse layers)
    "o1": (512, 512, 64, 1000.0 / 512, 9.0e-5, 200.0),
    # BASELINE config 5 scale: 184M-element two-level octree (134M fine + 50M coarse), 1 GPU
    "o2": (1024, 1024, 128, 1000.0 / 1024, 4.5e-5, 400.0),
}
OCT_COARSE_LAYERS = {"o1": 96, "o2": 192}
# Vp, Vs, rho of element (i, j, k) = the half-space's times a class factor in [0.9, 1.1]; class = hash(i, j, k) mod 61
LATERAL = {"c3h": (61, 0.1), "c2h": (61, 0.1), "m1h": (61, 0.1)}
# layered-basin models meshed by the Vs rule (hqh_layered_column) on several octree levels:
# name: (nx, ny, finest h [m], dt, freq, points per wavelength, coarsest cell [m], cells in depth,
#        [(ztop, vp, vs, rho), ...])
OCT_LAYERED = {
    # BASELINE config 5: a TeraShake-scale layered basin, 102.4 km x 102.4 km x 80 km, 0.5 Hz:
    # 100 m elements down to 16 km, then 200 / 400 / 800 m: 189M elements on four octree levels
    "o3": (1024, 1024, 100.0, 0.02, 0.5, 8, 800.0, 100,
           [(0.0, 1100.0, 600.0, 2000.0), (16000.0, 2000.0, 1100.0, 2300.0), (28800.0, 3600.0, 2000.0, 2500.0),
            (54400.0, 6000.0, 3464.0, 2700.0)]),
    "o3s": (256, 256, 100.0, 0.02, 0.5, 8, 800.0, 25,
            [(0.0, 1100.0, 600.0, 2000.0), (4000.0, 2000.0, 1100.0, 2300.0), (7200.0, 3600.0, 2000.0, 2500.0),
             (13600.0, 6000.0, 3464.0, 2700.0)]),
}
for _k, _v in OCT_LAYERED.items():
    WORKLOADS[_k] = (_v[0], _v[1], 0, _v[2], _v[3], _v[4])
# LATERALLY varying basin models meshed by hqh_octree_generate (the reference's Vs rule on setrec's 27-sample record +
# 2:1 balancing across faces and edges, pinned on tests/golden/c5_basin): a layered background and a sediment bowl whose
# depth varies with x and y, so the level interfaces have x-, y- and z-normal faces and staircase corners.
# name: domain (m), material grid cells along x, finest h (m), dt, freq, points per wavelength,
#       background [(ztop, vp, vs, rho)], bowl (xc, yc, a, b as fractions of Lx; depth in m; vp, vs, rho)
OCT_BASIN = {
    # BASELINE config 5 with lateral refinement: 102.4 km x 102.4 km x 51.2 km, 0.5 Hz, 100 m elements in the sediments
    "o4": dict(domain=(102400.0, 102400.0, 51200.0), grid=128, h=100.0, dt=0.02, freq=0.5, ppw=8,
               background=[(0.0, 2000.0, 1100.0, 2300.0), (4000.0, 3600.0, 2000.0, 2500.0), (12800.0, 6000.0, 3464.0, 2700.0)],
               bowl=(0.42, 0.55, 0.46, 0.40, 32000.0, 1100.0, 600.0, 2000.0)),
    "o4s": dict(domain=(25600.0, 25600.0, 12800.0), grid=64, h=100.0, dt=0.02, freq=0.5, ppw=8,
                background=[(0.0, 2000.0, 1100.0, 2300.0), (1600.0, 3600.0, 2000.0, 2500.0), (4800.0, 6000.0, 3464.0, 2700.0)],
                bowl=(0.42, 0.55, 0.40, 0.30, 6400.0, 1100.0, 600.0, 2000.0)),
    # The same two basins with a VELOCITY GRADIENT (round-5 review 5): Vp and Vs times f = 1 + gx x/Lx + gy y/Ly + gz z/Lz
    # (density times 1 + (f - 1) / 2) on a material grid of 200 m cells, so that no two neighbouring coarse elements share
    # (c1, c2, beta) -- what setrec's 27-sample average gives on any real CVM (psolve.c:1307-1397).  The per-element
    # kernels (hq_k_brick_het, element-form patches) then run INSIDE every octree level, beside the hanging nodes; no
    # assembled-stencil fast path applies anywhere.  f >= 1: the Vs rule picks the levels it picks for o4 / o4s.
    # Pinned in miniature on the real psolve: tests/golden/c5_gradient (make_cvm `grad`).
    "o4g": dict(domain=(102400.0, 102400.0, 51200.0), grid=512, h=100.0, dt=0.02, freq=0.5, ppw=8,
                background=[(0.0, 2000.0, 1100.0, 2300.0), (4000.0, 3600.0, 2000.0, 2500.0), (12800.0, 6000.0, 3464.0, 2700.0)],
                bowl=(0.42, 0.55, 0.46, 0.40, 32000.0, 1100.0, 600.0, 2000.0), gradient=(0.08, 0.05, 0.10)),
    "o4gs": dict(domain=(25600.0, 25600.0, 12800.0), grid=128, h=100.0, dt=0.02, freq=0.5, ppw=8,
                 background=[(0.0, 2000.0, 1100.0, 2300.0), (1600.0, 3600.0, 2000.0, 2500.0), (4800.0, 6000.0, 3464.0, 2700.0)],
                 bowl=(0.42, 0.55, 0.40, 0.30, 6400.0, 1100.0, 600.0, 2000.0), gradient=(0.08, 0.05, 0.10)),
}
for _k, _v in OCT_BASIN.items():
    WORKLOADS[_k] = (int(_v["domain"][0] / _v["h"]), int(_v["domain"][1] / _v["h"]), 0, _v["h"], _v["dt"], _v["freq"])
WORKLOAD_NAMES = {"c3": "64M-element uniform box 512x512x256, point double-couple source",
                  "c2": "8M-element uniform box 256x256x128, homogeneous half-space",
                  "c1": "examples/simple-sized box 16x16x8", "m1": "1M-element box 128x128x64",
                  "c3h": "64M-element box 512x512x256, Vp/Vs/rho of every element perturbed +-10 % (61 classes by a hash of its three indices), point double-couple source",
                  "c2h": "8M-element box 256x256x128, Vp/Vs/rho of every element perturbed +-10 %",
                  "m1h": "1M-element box 128x128x64, Vp/Vs/rho of every element perturbed +-10 %",
                  "o1": "23M-element two-level octree box (soft 64-layer top refined 2:1, 262k hanging nodes)",
                  "o2": "184M-element two-level octree box (1024x1024x128 fine over 512x512x192 coarse, 1M hanging nodes)",
                  "o3": "189M-element layered basin (102.4 km x 102.4 km x 80 km, 0.5 Hz) on four octree levels (100-800 m)",
                  "o3s": "3M-element layered basin on four octree levels (small version of o3)",
                  "o4": "laterally refined basin (102.4 km x 102.4 km x 51.2 km, 0.5 Hz): sediment bowl in a layered half-space, "
                        "octree levels of 100-800 m with x-, y- and z-normal interfaces (Vs rule + 2:1 balance as the reference's mesher)",
                  "o4s": "small laterally refined basin (25.6 km x 25.6 km x 12.8 km), four octree levels",
                  "o4g": "laterally refined basin with a velocity gradient (102.4 km x 102.4 km x 51.2 km, 0.5 Hz): the bowl and layers "
                         "of o4, Vp / Vs / rho varying on a 200 m grid so that every coarse element has its own (c1, c2, beta) -- "
                         "per-element-coefficient kernels inside every octree level, hanging nodes, no assembled-stencil fast path",
                  "o4gs": "small laterally refined basin with a velocity gradient (25.6 km x 25.6 km x 12.8 km), four octree levels"}


def usable_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(seconds_target=6.0, single=False):
    """The oracle's reference-formulation loops (effective stiffness + the 8x8
    conventional damping loop + nodal update) on the host cores: one independent
    64x64x32 partition per core, seeded random field (all elements active)."""
    from oracle import herc_oracle as ho
    ho.lib()
    cores = usable_cores()
    nx, ny, nz, h, dt, freq = 64, 64, 32, 1000.0 / 128, 3.6e-4, 50.0
    elem_ijk, lnid, node_ijk = ho.uniform_mesh(nx, ny, nz)
    edata = np.empty((len(lnid), 4), np.float32)
    edata[:] = (h, 6000.0, 3464.0, 2700.0)
    et, nt = ho.solver_init(lnid, edata, ho.face_bits(elem_ijk, nx, ny, nz), len(node_ijk), dt, freq)
    K = ho.compute_K()
    E, N = len(lnid), len(node_ijk)
    rng = np.random.default_rng(12345)
    base1 = rng.uniform(-1, 1, (N, 3)) * 1e-3
    base2 = base1 + rng.uniform(-1, 1, (N, 3)) * 1e-6

    def timed(formulation, steps):
        state = [(base1.copy(), base2.copy()) for _ in range(cores)]

        def work(i):
            ho.solver_run(lnid, et, nt, state[i][0], state[i][1], 0, steps, dt, formulation=formulation, K=K)
        th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        return time.perf_counter() - t0

    t1 = timed(0, 1)                                   # calibrate
    steps = int(max(2, min(200, seconds_target / max(t1, 1e-3))))
    tref = timed(0, steps)
    tf1 = timed(1, 2)
    fsteps = int(max(2, min(400, 3.0 / max(tf1 / 2, 1e-3))))
    tfused = timed(1, fsteps)
    port = {
        "value": cores * E * steps / tref,
        "unit": "element-updates/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d independent %dx%dx%d partitions (one per core), %d steps, reference formulation "
                  "(effective stiffness + conventional Rayleigh damping loop + nodal update), "
                  "random field, zero-skip on but no quiescent element" % (cores, nx, ny, nz, steps),
        "per_core": E * steps / tref,
        "fused_formulation_value": cores * E * fsteps / tfused,
    }
    # the REAL reference (MPI, one rank per core) when its prebuilt binary travelled with the repo
    if os.environ.get("HQ_BENCH_NO_REFERENCE"):
        return port
    try:
        from oracle import ref_baseline as rb
        if single:                       # --precision f32: the reference built with -DSINGLE_PRECISION_SOLVER beside it
            rb.PSOLVE = rb.PSOLVE + "_f32" if not rb.PSOLVE.endswith("_f32") else rb.PSOLVE
        if not rb.available():
            return port
        why_not_c2 = None
        try:
            # the C2 box, 256 x 256 x 128 = 8 388 608 elements (~11 GB over the ranks)
            r = rb.measure_box(cores, freq=80.0, dt=0.00025, steps=101, timeout=330)
        except Exception as exc:
            why_not_c2 = str(exc)[:160] or type(exc).__name__
            r = rb.measure_box(cores, freq=40.0, dt=0.0005, steps=101, timeout=240)
        out = {
            "value": r["value"],
            "unit": "element-updates/s",
            "cores": cores,
            "kind": "reference",
            "sample": "CMU-Quake/hercules psolve itself (oracle/_ref/" + os.path.basename(rb.PSOLVE) + ", MPICH, %d ranks): examples/simple "
                      "material refined by its own mesher to %d elements (f = %s Hz), effective stiffness + "
                      "Rayleigh damping, %d point sources so that no element is quiescent; the solver's own "
                      "wall clock over steps %d..%d of one run"
                      % (cores, r["elements"], "80" if why_not_c2 is None else "40", r["sources"], r["steps"][0], r["steps"][1]),
            "per_core": r["value"] / cores,
            "elements": int(r["elements"]),
            "s_per_step": r["s_per_step"],
            # the same program on the 64 M-element box the metric is quoted on: 2.55 s per step, too long for every run --
            # recorded once (profiles/r05/ref_psolve_64m.json, 16 ranks, steps 150..200, every element active)
            "recorded_64m": {"value": 26.3e6, "elements": 67108864, "cores": 16, "s_per_step": 2.55,
                             "file": "profiles/r05/ref_psolve_64m.json"},
            "port_value": port["value"],
            "port_fused_formulation_value": port["fused_formulation_value"],
        }
        if why_not_c2 is not None:
            out["c2_box_not_run"] = why_not_c2
        return out
    except Exception as exc:                            # mpiexec not usable on this box: keep the port
        port["reference_error"] = str(exc)[:200]
        return port


def seeded_field(node_ijk, nx, ny, interfaces=()):
    """Displacements as a function of the global node coordinates (so every copy of a shared node
    starts equal).  Octree boxes: interfaces = [(z of the plane, finer edge hf), ...] in finest
    units; a node of such a plane that is not on the coarser grid hangs, and starts at the mean of
    its anchors (its 2 edge or 4 face neighbours at +-hf, compute_adjust ASSIGNMENT)."""
    if isinstance(interfaces, dict):
        return interfaces["field"].copy()                   # OCT_BASIN: computed on the whole mesh by make_octbox
    ijk = np.asarray(node_ijk).astype(np.int64)

    def raw(i, j, k):
        gid = (k * (ny + 1) + j) * (nx + 1) + i
        out = np.empty((len(gid), 3))
        for d in range(3):
            x = (gid * 3 + d + 12345) * np.int64(2654435761) % np.int64(2 ** 31)
            out[:, d] = (x.astype(np.float64) / 2 ** 30 - 1.0) * 1e-3
        return out
    i, j, k = ijk[:, 0], ijk[:, 1], ijk[:, 2]
    u = raw(i, j, k)
    for z, hf in interfaces:
        on = k == z
        ox, oy = on & (i % (2 * hf) == hf), on & (j % (2 * hf) == hf)
        xe, ye, zf = ox & ~oy, oy & ~ox, ox & oy
        u[xe] = 0.5 * (raw(i[xe] - hf, j[xe], k[xe]) + raw(i[xe] + hf, j[xe], k[xe]))
        u[ye] = 0.5 * (raw(i[ye], j[ye] - hf, k[ye]) + raw(i[ye], j[ye] + hf, k[ye]))
        u[zf] = 0.25 * (raw(i[zf] - hf, j[zf] - hf, k[zf]) + raw(i[zf] + hf, j[zf] - hf, k[zf]) +
                        raw(i[zf] - hf, j[zf] + hf, k[zf]) + raw(i[zf] + hf, j[zf] + hf, k[zf]))
    return u


def basin_grid(spec):
    """The material model of an OCT_BASIN workload on its grid: vp, vs, rho [nz][ny][nx] float32 (mesh axes), cell edge."""
    Lx, Ly, Lz = spec["domain"]
    n = spec["grid"]
    cell = Lx / n
    ny, nz = int(round(Ly / cell)), int(round(Lz / cell))
    z, y, x = np.meshgrid((np.arange(nz) + 0.5) * cell, (np.arange(ny) + 0.5) * cell, (np.arange(n) + 0.5) * cell, indexing="ij")
    vp, vs, rho = [np.zeros(z.shape, np.float32) for _ in range(3)]
    for ztop, a, b, c in spec["background"]:
        sel = z >= ztop
        vp[sel], vs[sel], rho[sel] = a, b, c
    xc, yc, a, b, depth, bvp, bvs, brho = spec["bowl"]
    sed = z < depth * np.maximum(0.0, 1.0 - ((x - xc * Lx) / (a * Lx)) ** 2 - ((y - yc * Lx) / (b * Lx)) ** 2)
    vp[sed], vs[sed], rho[sed] = bvp, bvs, brho
    del sed
    if "gradient" in spec:
        gx, gy, gz = spec["gradient"]
        f = 1.0 + gx * x / Lx + gy * y / Ly + gz * z / Lz
        del x, y, z
        vp, vs = (vp * f).astype(np.float32), (vs * f).astype(np.float32)
        rho = (rho * (1.0 + (f - 1.0) / 2.0)).astype(np.float32)
    return vp, vs, rho, cell


def basin_leaves(workload):
    """-> (elem_ticks, elem_edge, edata, far_ticks, ticksize) of an OCT_BASIN workload (hqh_octree_generate)."""
    from hercules_amd import host as hhost
    spec = OCT_BASIN[workload]
    vp, vs, rho, cell = basin_grid(spec)
    return hhost.octree_generate(vp, vs, rho, cell, spec["domain"], spec["freq"] * spec["ppw"], 0.0)


def basin_field(box, nx, ny):
    """Seeded start field of a WHOLE octree mesh: a function of the node coordinates, hanging nodes at the mean of their
    anchors (compute_adjust ASSIGNMENT, psolve.c:5992-6035; anchors never hang in a mesh balanced across faces and edges)."""
    u = seeded_field(box.node_xyz, nx, ny)
    ids, ptr, anchors = box.dangling
    if len(ids):
        deps = np.diff(ptr).astype(np.int64)
        u[ids] = np.add.reduceat(u[anchors], ptr[:-1].astype(np.int64), axis=0) / deps[:, None]
    return u


_BASIN_CACHE = {}         # workload -> (leaves, whole-mesh start field, E, N): what every partition of one process shares


def make_octbox(workload, rank, nranks):
    """-> (OctBox, total elements, total nodes, interfaces for seeded_field)"""
    from hercules_amd import host as hhost
    nx, ny, nz, h, dt, freq = WORKLOADS[workload]
    if workload in OCT_BASIN:
        if nranks > 1 and workload in _BASIN_CACHE:
            (ticks, edge, edata, far), field, E, N = _BASIN_CACHE[workload]
        else:
            ticks, edge, edata, far, ticksize = basin_leaves(workload)
            assert abs(int(edge.min()) * ticksize - h) < 1e-9 * h, "the finest leaf is not the workload's h"
            whole = hhost.OctBox.from_leaves(ticks, edge, edata, far, dt, freq)
            field = basin_field(whole, nx, ny)
            E, N = whole.E, whole.N
            if nranks == 1:
                return whole, E, N, {"field": field}
            whole.close()
            if E < 8000000:               # the small basin: the leaves and the whole-mesh field serve every rank of this process
                _BASIN_CACHE[workload] = ((ticks, edge, edata, far), field, E, N)
        box = hhost.OctBox.from_leaves(ticks, edge, edata, far, dt, freq, rank=rank, nranks=nranks)
        return box, E, N, {"field": field[box.gid]}
    if workload in OCT_LAYERED:
        _, _, _, _, _, ppw, h0, ncoarse, model = OCT_LAYERED[workload]
        col = hhost.layered_column(model, h0, ncoarse, freq * ppw)
        hf, levels = hhost.levels_from_column(col)
        assert hf == h
        box = hhost.OctBox(nx, ny, 0, 0, h, dt, freq, levels=levels, rank=rank, nranks=nranks)
    else:
        nzc = OCT_COARSE_LAYERS[workload]
        levels = [(nz, None), (nzc, None)]
        box = hhost.OctBox(nx, ny, nz, nzc, h, dt, freq, rank=rank, nranks=nranks)
    E, N, interfaces = octbox_counts(nx, ny, [l[0] for l in levels])
    return box, E, N, interfaces


def make_octbox_interfaces(workload):
    """The level interfaces of an octree workload for seeded_field (what make_octbox returns as its fourth value)."""
    from hercules_amd import host as hhost
    nx, ny, nz, h, dt, freq = WORKLOADS[workload]
    assert workload not in OCT_BASIN, "basins keep their start field (make_octbox's fourth value)"
    if workload in OCT_LAYERED:
        _, _, _, _, _, ppw, h0, ncoarse, model = OCT_LAYERED[workload]
        _, levels = hhost.levels_from_column(hhost.layered_column(model, h0, ncoarse, freq * ppw))
        layers = [l[0] for l in levels]
    else:
        layers = [nz, OCT_COARSE_LAYERS[workload]]
    return octbox_counts(nx, ny, layers)[2]


def octbox_counts(nx, ny, layers):
    """Elements, nodes and level interfaces [(z of the plane, finer edge) in finest-element units] of a layered octree
    box with layers[L] element layers of edge 2^L."""
    E = N = z = 0
    interfaces = []
    for L, n in enumerate(layers):
        E += (nx >> L) * (ny >> L) * n
        N += ((nx >> L) + 1) * ((ny >> L) + 1) * n
        if L > 0:
            interfaces.append((z, 1 << (L - 1)))
        z += n << L
    N += (nx + 1) * (ny + 1)
    return E, N, interfaces


def inproc_diagnostic(args):
    # HIP multiplexes a process's streams onto 4 hardware queues by default: eight partitions' streams then wait for each
    # other two by two.  One queue per partition (measured, 8 partitions of the 64M box: 1.77 -> 1.61 ms per step).
    # (Not with the chain on a second stream per partition: 16-24 streams with cross-stream events on 16-32 hardware
    # queues took 1.7 SECONDS per step -- the runtime resolves those waits on the host.)
    if os.environ.get("HQ_OVERLAP", "0") in ("", "0"):
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import hercules_amd as ha
    from hercules_amd import capi, host as hhost
    nx, ny, nz, h, dt, freq = WORKLOADS[args.workload]
    P = args.inproc_parts
    variant = {"auto": ha.HQ_VARIANT_AUTO, "scatter": ha.HQ_VARIANT_SCATTER, "patch": ha.HQ_VARIANT_PATCH}[args.variant]
    boxes, solvers = [], []
    octree = args.workload in OCT_COARSE_LAYERS or args.workload in OCT_LAYERED or args.workload in OCT_BASIN
    for r in range(P):
        if octree:
            b, total_e, _, interfaces = make_octbox(args.workload, r, P)
            sch = b.schedules()
            peers = set(q for kind in sch.values() for lst in kind.values() for q, _ in lst)
            b.info = {"total_elements": total_e,
                      "shared_nodes": int(sum(len(m) for kind in sch.values() for lst in kind.values() for _, m in lst)),
                      "nneighbors": len(peers)}
            u1 = seeded_field(b.node_xyz, nx, ny, interfaces)
        else:
            ncls, amp = LATERAL.get(args.workload, (0, 0.0))
            b = hhost.Box(nx, ny, nz, h, dt, freq, rank=r, nranks=P, lateral_classes=ncls, lateral_amp=amp)
            u1 = seeded_field(b.node_ijk, nx, ny)
        solvers.append(b.create_solver(variant=variant, tm1=u1, tm2=u1 * (1.0 - 1e-3), precision=args.precision))
        boxes.append(b)
    capi.group_link(solvers)
    capi.group_run(solvers, args.warmup)
    t0 = time.perf_counter()
    capi.group_run(solvers, args.steps)
    el = time.perf_counter() - t0
    E = boxes[0].info["total_elements"]
    print(json.dumps({"diagnostic": "in-process partitions on one GPU", "parts": P, "workload": args.workload,
                      "value": E * args.steps / el, "ms_per_step": el / args.steps * 1e3,
                      "shared_nodes": [b.info["shared_nodes"] for b in boxes],
                      "neighbors": [b.info["nneighbors"] for b in boxes],
                      "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "overlap": os.environ.get("HQ_OVERLAP", "0")}))


def select_transport(want, solver, new_solver, bring_up, trial, trials, agree=None):
    """Which transport carries the halo records of an N > 1 run.  Pure control flow (tests/test_bench_roofline_cpu.py
    drives it with stand-ins): `solver` is the context built so far, `new_solver()` builds another one on the same mesh,
    `bring_up(kind, solver)` -> True if transport `kind` is in place on EVERY rank, `trial(solver)` -> ms per step or None.
    want = auto: RCCL and IPC are both brought up (a context each), both timed, the faster kept; one of them alone if the
    other does not come up; host-staged if neither does.  `agree(a, b)` (optional) -> do the two timed contexts hold the
    same field after their identical trial runs?  If not, the IPC transport -- peer stores ordered by flags, the newer
    and less proven of the two -- is rejected whatever it measured (round-4 advisor).  want = rccl / ipc: that one, else the other device-side one
    (rccl -> ipc), else host-staged.  A context whose bring-up failed is closed, never reused.
    -> (solver that carries the chosen transport, its name); `trials` is filled with what was measured."""
    chosen = None
    if want == "auto":
        up = {}
        for kind in ("rccl", "ipc"):
            sv = solver if kind == "rccl" else new_solver()
            if bring_up(kind, sv):
                up[kind] = sv
            else:
                sv.close()
        if len(up) == 2:
            for kind, sv in up.items():
                trials[kind] = trial(sv)
            alive = {k: v for k, v in trials.items() if v is not None}
            chosen = min(alive, key=alive.get) if alive else None
            if agree is not None and len(alive) == 2 and not agree(up["rccl"], up["ipc"]):
                trials["ipc_rejected"] = "its field differs from the RCCL context's after the same trial run"
                chosen = "rccl"
        elif up:
            chosen = next(iter(up))
        for kind, sv in up.items():
            if kind != chosen:
                sv.close()
        if chosen:
            return up[chosen], chosen
        solver = new_solver()
        want = "host"
    if want in ("rccl", "ipc"):
        if bring_up(want, solver):
            return solver, want
        solver.close()
        if want == "rccl":
            solver = new_solver()
            if bring_up("ipc", solver):
                return solver, "ipc"
            solver.close()
        solver = new_solver()
    bring_up("host", solver)
    return solver, "host"


def flush_c_stdio():
    """fflush(NULL): libraries that printf (RCCL's banner) must not leave text in a C buffer that
    would reach stdout after this script's one JSON line."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: this parent starts N rank processes itself --
    fresh children, one GPU each (LOCAL_RANK), rendezvous on 127.0.0.1 -- relays rank 0's JSON line
    and exits with the worst child status.  The parent makes no GPU call and never re-execs
    (the reference's counterpart is `mpiexec -np N psolve`, psolve.c:7344-7389)."""
    n = args.gpus
    if not args.dry_launch:
        import torch                                       # device_count() does not initialise the GPU
        have = torch.cuda.device_count()
        if have < n and not (have >= 1 and os.environ.get("HQ_BENCH_SHARE_GPU")):
            print("bench.py: --gpus %d but this box shows %d GPU(s); nothing launched" % (n, have), file=sys.stderr)
            return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   HQ_BENCH_LAUNCHED_BY_PARENT="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      universal_newlines=True))
    # rank 0's stdout is read by a thread; the parent polls ALL children: a rank that dies during set-up or rendezvous
    # would otherwise leave the others (and this parent) in the rendezvous' 30-minute timeout.  stderr of every rank
    # is inherited, so the cause is on the parent's stderr.
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = r
        time.sleep(0.2)
    if failed is not None:
        print("bench.py: rank %d exited with status %d; stopping the other ranks" % (failed, procs[failed].returncode),
              file=sys.stderr)
        for p in procs:
            if p.poll() is None:
                p.terminate()                              # the exact children started above, nothing else
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    out0 = chunks[0] if chunks else ""
    line = [l for l in (out0 or "").splitlines() if l.startswith("{")]
    for l in (out0 or "").splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if line:
        print(line[-1], flush=True)
    return max(abs(rc) for rc in rcs) if any(rcs) else (0 if line else 1)


def dry_launch_worker(args, rank, world):
    """--dry-launch: the ranks only rendezvous (gloo, CPU) and count each other; no GPU, no solver."""
    import torch
    import torch.distributed as dist
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    t = torch.tensor([1.0, float(rank)], dtype=torch.float64)
    dist.all_reduce(t)
    ok = int(t[0]) == world and int(t[1]) == world * (world - 1) // 2
    dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "ranks_seen": int(t[0]), "ok": ok,
                          "launched_by": "bench.py" if os.environ.get("HQ_BENCH_LAUNCHED_BY_PARENT") else "external launcher"}),
              flush=True)
    dist.destroy_process_group()
    return 0 if ok else 1


def build_problem(args, rank, world, device):
    """(box, solver, N, octree) of the workload, field and source in place -- shared by the timed
    run and the counter-collection child."""
    import hercules_amd as ha
    from hercules_amd import host as hhost
    nx, ny, nz, h, dt, freq = WORKLOADS[args.workload]
    octree = args.workload in OCT_COARSE_LAYERS or args.workload in OCT_LAYERED or args.workload in OCT_BASIN
    variant = {"auto": ha.HQ_VARIANT_AUTO, "scatter": ha.HQ_VARIANT_SCATTER, "patch": ha.HQ_VARIANT_PATCH}[args.variant]
    interfaces = ()
    if octree:
        box, total_e, total_n, interfaces = make_octbox(args.workload, rank, world)
        box.info = {"nharbored": box.N, "total_elements": total_e, "lenum": box.E, "total_nodes": total_n}
        box.node_ijk = box.node_xyz
        box.start_interfaces = interfaces
    else:
        ncls, amp = LATERAL.get(args.workload, (0, 0.0))
        box = hhost.Box(nx, ny, nz, h, dt, freq, rank=rank, nranks=world, lateral_classes=ncls, lateral_amp=amp)
    N = box.info["nharbored"]
    # seeded random start: identical on every rank for shared nodes
    # because it is a function of the global node coordinates
    u1 = seeded_field(box.node_ijk, nx, ny, interfaces)
    u2 = u1 * (1.0 - 1e-3)
    solver = box.create_solver(variant=variant, device=device, tm1=u1, tm2=u2, precision=args.precision, options=solver_options(world))
    del u1, u2
    return box, solver, N, octree


def solver_options(world):
    """Typed options of every context this script builds (hq_options; nothing through the environment -- experiments set
    HQ_ALLOW_ENV=1 and HQ_* themselves): between ranks a wait for a neighbour gives up after 5 s instead of 20."""
    return {"ipc_timeout_ms": 5000.0} if world > 1 else None


def add_source(args, box, solver, octree, total_steps):
    nx, ny, nz, h, dt, freq = WORKLOADS[args.workload]
    L = nx * h
    if not octree:
        loaded, pattern = box.point_source(L / 2, L / 2, L / 5, 0.0, 90.0, 0.0)
        rp = box.run_params(loaded=loaded, pattern=pattern, moment=1e12, rise_time=20 * dt,
                            source_window=max(total_steps, 1))
        if len(loaded):
            solver.set_source(loaded, box.source_table(rp, 0, total_steps), 0)


def pmc_child(args):
    """Run under `rocprofv3 --pmc ...` by measure_traffic(): the same workload, a few steps, nothing
    printed.  No torch here: the C-ABI alone (device 0)."""
    os.environ["OMP_NUM_THREADS"] = str(usable_cores())
    box, solver, N, octree = build_problem(args, 0, 1, 0)
    add_source(args, box, solver, octree, args.warmup + args.steps)
    solver.run(args.warmup + args.steps)
    solver.sync()
    solver.close()
    box.close()
    return 0


def measure_traffic(args, keep_dir=None):
    """HBM bytes per launch of the dominant kernel, measured in THIS session: two separate
    `rocprofv3 --pmc` passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with
    tracing) over a short run of the same workload, corrected as MI355X_MICROARCH.md (HBM) prescribes
    for gfx950: FETCH_SIZE x 2 for wide streaming reads, both counters in KiB.  The passes run as child
    processes BEFORE this process touches the GPU.  -> dict or {"error": ...}"""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    out_root = os.path.abspath(keep_dir) if keep_dir else tempfile.mkdtemp(prefix="hq_pmc_", dir="/tmp")
    os.makedirs(out_root, exist_ok=True)
    res = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out_root, counter.lower())
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--",
                   os.path.realpath(sys.executable), os.path.abspath(__file__), "--pmc-child",
                   "--workload", args.workload, "--variant", args.variant, "--precision", args.precision,
                   "--steps", str(PMC_STEPS - 1), "--warmup", "1"]
            env = dict(os.environ, TMPDIR="/tmp")
            try:
                p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                   universal_newlines=True, timeout=420)
            except subprocess.TimeoutExpired:
                return {"error": "%s pass timed out" % counter}
            if p.returncode != 0:
                return {"error": "%s pass failed (rc %d): %s" % (counter, p.returncode, (p.stdout or "")[-300:])}
            per_kernel = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get("Counter_Name") != counter:
                        continue
                    k = r.get("Kernel_Name", "").split("(")[0]
                    a = per_kernel.setdefault(k, [0, 0.0])
                    a[0] += 1
                    a[1] += float(r["Counter_Value"])
            if not per_kernel:
                return {"error": "%s pass wrote no counter rows" % counter}
            res[counter] = per_kernel
    finally:
        if keep_dir is None:
            shutil.rmtree(out_root, ignore_errors=True)
    return res


PMC_STEPS = 4              # steps a counter pass runs (pmc_child: 1 + 3)
STEP_KERNELS = ("hq_k_brick", "hq_k_patch", "hq_k_element", "hq_k_update")


def traffic_of(pmc, kernel, steps=PMC_STEPS):
    """(HBM bytes per step of the step's compute kernels, corrected read bytes, write bytes, steps profiled).
    A step launches the dominant kernel (once, or two to three times: hq_k_patch_stencil has a launch per patch
    class, hq_k_brick one per n_t form) and the patch kernels for the nodes it does not take (domain faces,
    hanging nodes): `kernel_ms` spans them all, so do the bytes.  `steps` is the number of steps the counter
    pass RAN (known from its command line) -- never the dispatch count of some kernel."""
    if not any(kernel in k for k in pmc["FETCH_SIZE"]) or not any(kernel in k for k in pmc["WRITE_SIZE"]):
        return None

    def per_step(counter):
        return sum(t for k, (n, t) in pmc[counter].items() if any(name in k for name in STEP_KERNELS)) / steps
    f, w = per_step("FETCH_SIZE"), per_step("WRITE_SIZE")
    rd, wr = f * 1024.0 * 2.0, w * 1024.0
    return rd + wr, rd, wr, steps


PARITY_STEPS = 3                # steps of the parity run behind the timed region
PARITY_EDGE = 16                # elements per axis of a window (a power of two: hqh_box_create)
PARITY_TOL = 1e-9               # the GPU parity bar: relative L-inf on nodal displacement
PARITY_TOL_F32 = 2e-6           # --precision f32: a float state against the oracle's float build over a few steps
                                # (tests/test_gpu_single_precision.py states where the figure comes from)


def parity_tol(args):
    return PARITY_TOL_F32 if getattr(args, "precision", "f64") == "f32" else PARITY_TOL
PARITY_WORKLOADS = ("c1", "c2", "c3", "m1", "c3h", "c2h", "m1h")     # boxes: a window of them is a small box of the same
                                                                     # material (hqh_box_params.origin: classes at the big box's indices)


def parity_windows(args, box, solver, rank, world):
    """Parity that travels with the bench line (round-4 review, item 5): BEHIND the timed region the context -- same
    partition, same transport -- is reset to the seeded start field and stepped PARITY_STEPS times; after k steps a node
    depends on its k-ring only, so a window of 16^3 elements cut from the box (here: built as a box of its own with the
    same h, dt and material, domain faces where the window touches the domain's) and stepped by the oracle's reference
    loops gives the exact values of the nodes at least k layers inside every cut face.  The windows are centred on nodes
    this rank SHARES with others (its partition interfaces: pack, transport, interface update and unpack are inside the
    checked cones), plus brick-tile borders and domain faces.
    -> (windows, nodes checked, worst relative error) of this rank; None for workloads without a window oracle."""
    single = getattr(args, "precision", "f64") == "f32"
    real = np.float32 if single else np.float64
    if args.workload in OCT_COARSE_LAYERS or args.workload in OCT_LAYERED or args.workload in OCT_BASIN:
        return parity_windows_octree(args, box, solver) if world == 1 else parity_windows_octree_partitioned(args, box, solver, rank, world)
    if args.workload not in PARITY_WORKLOADS:
        return None
    from hercules_amd import host as hhost
    from oracle import herc_oracle as ho             # the checker, behind the timed region only
    nx, ny, nz, h, dt, freq = WORKLOADS[args.workload]
    k, W = PARITY_STEPS, PARITY_EDGE
    dims = (nx, ny, nz)
    if min(dims) < W:
        return None
    u1 = seeded_field(box.node_ijk, nx, ny)
    solver.set_source(np.zeros(0, np.int32), np.zeros((0, 0, 3)))
    solver.upload(u1, u1 * (1.0 - 1e-3), 0)
    solver.run(k)
    solver.sync()
    ijk = box.node_ijk.astype(np.int64)
    key = (ijk[:, 2] * (ny + 1) + ijk[:, 1]) * (nx + 1) + ijk[:, 0]
    order = np.argsort(key)
    skey = key[order]
    centres = []
    if world > 1:
        sch = box.schedule()
        shared = np.unique(np.concatenate([m for _, m in sch["c"] + sch["s"]] or [np.zeros(0, np.int32)]))
        if len(shared):
            centres += [tuple(ijk[n]) for n in shared[np.linspace(0, len(shared) - 1, 12).astype(np.int64)]]
    lo_n, hi_n = ijk.min(axis=0), ijk.max(axis=0)
    mid = (lo_n + hi_n) // 2
    centres += [tuple(mid), (int(lo_n[0]) + 64, int(lo_n[1]) + 8, int(mid[2])), (int(lo_n[0]), int(mid[1]), int(mid[2])),
                (int(mid[0]), int(mid[1]), int(lo_n[2])), (int(hi_n[0]), int(hi_n[1]), int(hi_n[2]))]
    scale = np.abs(u1).max()
    nwin, nchecked, worst, seen = 0, 0, 0.0, set()
    for c in centres:
        lo = tuple(int(min(max(c[d] - W // 2, 0), dims[d] - W)) for d in range(3))
        if lo in seen:
            continue
        seen.add(lo)
        ncls, amp = LATERAL.get(args.workload, (0, 0.0))
        sub = hhost.Box(W, W, W, h, dt, freq, lateral_classes=ncls, lateral_amp=amp, origin=lo)
        g = sub.node_ijk.astype(np.int64) + np.array(lo)
        w1 = seeded_field(g, nx, ny)
        # (f32: the oracle's float build on the tables the context was handed -- the n_t rows rounded to float)
        o1, o2 = (w1 * (1.0 - 1e-3)).astype(real), w1.astype(real)
        ho.solver_run(sub.lnid, sub.etable.copy(), np.ascontiguousarray(sub.ntable, real), o1, o2, 0, k, dt)
        ok = np.ones(len(g), bool)
        for d in range(3):
            if lo[d] > 0:                                   # a cut face unless it is the domain's own near face
                ok &= g[:, d] >= lo[d] + k
            if lo[d] + W < dims[d]:
                ok &= g[:, d] <= lo[d] + W - k
        sub.close()
        gk = (g[:, 2] * (ny + 1) + g[:, 1]) * (nx + 1) + g[:, 0]
        pos = np.searchsorted(skey, gk)
        pos[pos >= len(skey)] = 0
        mine = ok & (skey[pos] == gk)                       # the checked nodes this rank harbors
        if not mine.any():
            continue
        tm1, tm2 = solver.gather(order[pos[mine]].astype(np.int32))
        worst = max(worst, float(np.abs(tm1.astype(np.float64) - o2[mine]).max() / scale),
                    float(np.abs(tm2.astype(np.float64) - o1[mine]).max() / scale))
        nwin += 1
        nchecked += int(mine.sum())
    return nwin, nchecked, worst


PARITY_OCTREE_WINDOWS = 6      # windows of an octree workload (each costs a pass over the mesh to cut out)


def parity_windows_octree(args, box, solver):
    """The same for the octree workloads on ONE rank, where the whole mesh is at hand: windows centred on hanging nodes
    of different kinds (orientation x level pair; oracle/windows.py, pinned in tests/test_octree_windows_cpu.py), cut out
    with the true table rows, stepped by the oracle with compute_adjust.  Partitions of an octree mesh carry no windows
    (a window across a partition interface needs the neighbour's rows): the partitioned octree paths are pinned by the
    tests against the single-partition runs and the reference's own stripes."""
    from oracle import windows as ow                 # the checker, behind the timed region only
    nx, ny = WORKLOADS[args.workload][:2]
    k = 2
    if box.ldnnum == 0:
        return None
    u1 = seeded_field(box.node_ijk, nx, ny, box.start_interfaces)
    solver.set_source(np.zeros(0, np.int32), np.zeros((0, 0, 3)))
    solver.upload(u1, u1 * (1.0 - 1e-3), 0)
    solver.run(k)
    solver.sync()
    xyz = box.node_xyz
    elem_lo = xyz[box.lnid[:, 0]].astype(np.int32)
    elem_edge = xyz[box.lnid[:, 1], 0] - elem_lo[:, 0]
    deps, mask, dist = ow.hanging_kinds(xyz, box.dangling)
    kinds = sorted(set(zip(mask.tolist(), dist.tolist())))
    pick = [kinds[i] for i in np.unique(np.linspace(0, len(kinds) - 1, PARITY_OCTREE_WINDOWS).astype(int))]
    scale = np.abs(u1).max()
    # (--precision f32: the oracle's float build on the values and rows the context was handed -- rounded to float)
    real = np.float32 if getattr(args, "precision", "f64") == "f32" else np.float64
    w1, w2 = u1.astype(real), (u1 * (1.0 - 1e-3)).astype(real)
    ntab = np.ascontiguousarray(box.ntable, real)
    nwin, nchecked, worst = 0, 0, 0.0
    for lo, hi, margin, centre, cand in ow.lateral_windows(xyz, box.dangling, elem_lo, elem_edge, k, per_kind=1, kinds=set(pick)):
        win = ow.octree_window(box.lnid, xyz, box.dangling, elem_lo, elem_edge, lo, hi, margin, cand)
        g1, g2 = ow.octree_window_oracle(win, box.etable, ntab, w1, w2, k, box.dt)
        ok, nodes = win["ok"], win["nodes"]
        tm1, tm2 = solver.gather(nodes[ok])
        worst = max(worst, float(np.abs(tm1.astype(np.float64) - g1[ok]).max() / scale),
                    float(np.abs(tm2.astype(np.float64) - g2[ok]).max() / scale))
        nwin += 1
        nchecked += int(ok.sum())
    return nwin, nchecked, worst


def node_keys(xyz):
    """One int64 per node from its coordinates (finest-element units, < 2^21 per axis): equal on every rank that harbors it."""
    q = np.asarray(xyz).astype(np.int64)
    return q[:, 0] | (q[:, 1] << 21) | (q[:, 2] << 42)


def octree_window_records(args, whole, interfaces, k, prefer_keys=None, real=np.float64):
    """Rank 0's share of the partitioned octree parity (and a CPU-testable unit): on the WHOLE mesh, windows centred on
    hanging nodes of different kinds -- those that are shared between partitions first (prefer_keys: node_keys of the
    nodes the ranks' schedules name) --, stepped by the oracle.  -> [(keys of the checked nodes, tm1, tm2)], scale."""
    from oracle import windows as ow
    nx, ny = WORKLOADS[args.workload][:2]
    whole.node_ijk = whole.node_xyz
    u1 = seeded_field(whole.node_xyz, nx, ny, interfaces)
    xyz = whole.node_xyz
    elem_lo = xyz[whole.lnid[:, 0]].astype(np.int32)
    elem_edge = xyz[whole.lnid[:, 1], 0] - elem_lo[:, 0]
    dangling = whole.dangling
    if prefer_keys is not None and len(prefer_keys):
        # the hanging nodes the ranks share: windows around them hold a partition interface AND a level interface
        ids, ptr, anchors = dangling
        sel = np.nonzero(np.isin(node_keys(xyz[ids]), prefer_keys))[0]
        if len(sel) >= PARITY_OCTREE_WINDOWS:
            nptr = np.zeros(len(sel) + 1, np.int64)
            nptr[1:] = np.cumsum(ptr[sel + 1] - ptr[sel])
            nanc = np.concatenate([anchors[ptr[i]:ptr[i + 1]] for i in sel])
            pick_dn = (ids[sel], nptr.astype(np.int32), nanc)
        else:
            pick_dn = dangling
    else:
        pick_dn = dangling
    deps, mask, dist = ow.hanging_kinds(xyz, pick_dn)
    kinds = sorted(set(zip(mask.tolist(), dist.tolist())))
    pick = [kinds[i] for i in np.unique(np.linspace(0, len(kinds) - 1, PARITY_OCTREE_WINDOWS).astype(int))]
    w1, w2 = u1.astype(real), (u1 * (1.0 - 1e-3)).astype(real)
    ntab = np.ascontiguousarray(whole.ntable, real)
    records = []
    for lo, hi, margin, centre, cand in ow.lateral_windows(xyz, pick_dn, elem_lo, elem_edge, k, per_kind=1, kinds=set(pick)):
        win = ow.octree_window(whole.lnid, xyz, dangling, elem_lo, elem_edge, lo, hi, margin, cand)
        g1, g2 = ow.octree_window_oracle(win, whole.etable, ntab, w1, w2, k, whole.dt)
        ok, nodes = win["ok"], win["nodes"]
        records.append((node_keys(xyz[nodes[ok]]), g1[ok].astype(np.float64), g2[ok].astype(np.float64)))
    return records, float(np.abs(u1).max())


def parity_windows_octree_partitioned(args, box, solver, rank, world):
    """Octree workloads on N > 1 ranks (round-5 review 2a): every rank resets its context -- its partition, the transport
    that was timed -- to the start field and steps it twice; rank 0 ALSO builds the whole mesh (host only), cuts windows
    around hanging nodes that the partitions share, steps them with the oracle and broadcasts (node key, expected value)
    records; every rank checks the records of the nodes it harbors against what its context holds.
    -> (windows, nodes checked, worst relative error) of this rank."""
    import torch.distributed as dist
    nx, ny = WORKLOADS[args.workload][:2]
    k = 2
    u1 = seeded_field(box.node_ijk, nx, ny, box.start_interfaces)
    solver.set_source(np.zeros(0, np.int32), np.zeros((0, 0, 3)))
    solver.upload(u1, u1 * (1.0 - 1e-3), 0)
    solver.run(k)
    solver.sync()
    sch = box.schedules()
    shared = np.unique(np.concatenate([m for kind in sch.values() for lst in kind.values() for _, m in lst] or [np.zeros(0, np.int32)]))
    gathered = [None] * dist.get_world_size() if rank == 0 else None
    dist.gather_object(node_keys(box.node_xyz[shared]) if len(shared) else np.zeros(0, np.int64), gathered, dst=0)
    payload = [None]
    if rank == 0:
        real = np.float32 if getattr(args, "precision", "f64") == "f32" else np.float64
        whole, _, _, interfaces = make_octbox(args.workload, 0, 1)
        try:
            payload[0] = octree_window_records(args, whole, interfaces, k, np.unique(np.concatenate([g for g in gathered if g is not None])), real)
        finally:
            whole.close()
    dist.broadcast_object_list(payload, src=0)
    records, scale = payload[0]
    mykeys = node_keys(box.node_xyz)
    order = np.argsort(mykeys)
    skeys = mykeys[order]
    nwin, nchecked, worst = 0, 0, 0.0
    for keys, g1, g2 in records:
        pos = np.minimum(np.searchsorted(skeys, keys), len(skeys) - 1)
        mine = skeys[pos] == keys
        if not mine.any():
            continue
        tm1, tm2 = solver.gather(order[pos[mine]].astype(np.int32))
        worst = max(worst, float(np.abs(tm1.astype(np.float64) - g1[mine]).max() / scale),
                    float(np.abs(tm2.astype(np.float64) - g2[mine]).max() / scale))
        nwin += 1
        nchecked += int(mine.sum())
    return nwin, nchecked, worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=5,
                    help="back-to-back timed blocks of --steps steps, each bracketed by barrier + synchronize; the line "
                         "reports the MEDIAN block (ms_per_step, value) and every block in config.ms_per_step_runs")
    ap.add_argument("--preheat", type=float, default=0.3,
                    help="seconds of read-only device work (hq_check_finite) before the warm-up steps; 0 = none")
    ap.add_argument("--workload", default=os.environ.get("HQ_BENCH_WORKLOAD", "c3"), choices=sorted(WORKLOADS))
    ap.add_argument("--variant", default="auto", choices=["auto", "scatter", "patch"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the two rocprofv3 --pmc passes that measure roofline.traffic (N = 1 only)")
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"],
                    help="f32: libhq_solver_f32.so (hq_real = float, the reference's -DSINGLE_PRECISION_SOLVER) -- a separately "
                         "named dtype with its own oracle build and tolerance; never the headline")
    ap.add_argument("--pmc-dir", default=None, help="keep the rocprofv3 counter CSVs of the traffic passes here")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the oracle cone windows behind the timed region (config.parity_*)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dry-launch", action="store_true",
                    help="with --gpus N: start the N ranks, let them rendezvous over gloo on the CPU and exit "
                         "(launcher check, no GPU)")
    ap.add_argument("--inproc-parts", type=int, default=0,
                    help="diagnostic: P block partitions stepped in ONE process on ONE GPU with the "
                         "in-process halo transport (measures the cost of partitioning, not xGMI)")
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)
    if args.inproc_parts > 1:
        return inproc_diagnostic(args)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: launcher started %d rank(s), --gpus says %d: using the launcher's count" % (world, args.gpus),
                  file=sys.stderr)
        args.gpus = world
    if args.dry_launch:
        return dry_launch_worker(args, rank, world)

    # torchrun pins OMP_NUM_THREADS=1; the C host side builds the partition with OpenMP
    os.environ["OMP_NUM_THREADS"] = str(max(1, usable_cores() // max(world, 1)))
    # the host driver of this pool exports memory between processes through dmabuf only (hipIpcGetMemHandle fails without
    # this); the image exports it already -- a launcher that scrubs the environment must not cost the IPC transport
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    # Before this process touches the GPU: the traffic passes (child processes under rocprofv3) and
    # the CPU baseline (threads / MPI ranks on the host cores).  N = 1 only.
    pmc = None
    if rank == 0 and world == 1 and not args.no_pmc:
        pmc = measure_traffic(args, args.pmc_dir)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(single=args.precision == "f32")

    import torch
    import torch.distributed as dist
    import hercules_amd as ha

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if world > 1 and torch.cuda.device_count() < world and not os.environ.get("HQ_BENCH_SHARE_GPU"):
        raise SystemExit("bench.py: %d ranks but %d GPU(s) visible (one rank per GPU)" % (world, torch.cuda.device_count()))
    device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)

    t_setup = time.perf_counter()
    box, solver, N, octree = build_problem(args, rank, world, device)
    rccl_ranks = 1
    transport = "none (one rank)"

    def gloo_exchange(recvs, sends, tag):
        reqs = [dist.irecv(torch.from_numpy(buf), src=int(peer), tag=int(tag)) for peer, buf in recvs]
        reqs += [dist.isend(torch.from_numpy(buf), dst=int(peer), tag=int(tag)) for peer, buf in sends]
        for r in reqs:
            r.wait()
    # Transport between the ranks (HQ_BENCH_TRANSPORT = auto | rccl | ipc | host; default auto).
    #   rccl  RCCL grouped send/recv over xGMI (hq_comm_init)
    #   ipc   the engine's IPC transport: peer stores into IPC-exported receive buffers + epoch flags (hq_comm_init_ipc);
    #         also what ranks that share a GPU use (RCCL refuses duplicate devices)
    #   host  pinned host buffers + gloo (hq_comm_init_host): the last resort
    #   auto  bring up RCCL and IPC on a solver context each, time 10 steps of both, keep the faster one -- neither has
    #         ever met more than one GPU before the driver's scaling run, and their kernels differ in what they need
    #         beside the interior launch (RCCL's send/recv kernels want registers the brick workgroups hold; the IPC
    #         transport's pack / update / unpack kernels fit beside them).  A transport counts only if it
    #         came up on EVERY rank; config.transport names the one that ran, config.transport_trials what was measured.
    want = os.environ.get("HQ_BENCH_TRANSPORT", "auto") if world > 1 else "none"
    if world > 1 and torch.cuda.device_count() < world and want == "rccl":
        want = "ipc"                                       # ranks share a device: RCCL refuses duplicate devices
    errors, trials = {}, {}

    def everywhere(ok):
        t_ok = torch.tensor([float(ok)], dtype=torch.float64)
        dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
        return float(t_ok[0]) >= 1.0

    def start_field():
        nx_, ny_ = WORKLOADS[args.workload][:2]
        interfaces = () if not octree else (box.start_interfaces if args.workload in OCT_BASIN
                                            else make_octbox_interfaces(args.workload))
        return seeded_field(box.node_ijk, nx_, ny_, interfaces)

    def new_solver():
        """A second context on the same partition -- on EVERY rank or on none: a rank that cannot build one (device
        memory) must not leave the others waiting in the next collective (round-4 advisor)."""
        variant = {"auto": ha.HQ_VARIANT_AUTO, "scatter": ha.HQ_VARIANT_SCATTER, "patch": ha.HQ_VARIANT_PATCH}[args.variant]
        sv, why = None, ""
        try:
            u1 = start_field()
            sv = box.create_solver(variant=variant, device=device, tm1=u1, tm2=u1 * (1.0 - 1e-3), precision=args.precision,
                                   options=solver_options(world))
        except Exception as exc:
            why = str(exc)
        if not everywhere(sv is not None):
            if sv is not None:
                sv.close()
            raise SystemExit("bench.py: rank %d: a rank could not build a second solver context %s" % (rank, why))
        return sv

    def bring_up(kind, sv):
        """-> True if `kind` is in place on every rank's `sv`."""
        ok = 1
        if kind == "rccl":
            try:
                idbuf = [ha.capi.comm_unique_id() if rank == 0 else None]
            except ha.HqError as e:
                idbuf, ok, errors[kind] = [None], 0, str(e)
            dist.broadcast_object_list(idbuf, src=0)
            if idbuf[0] is not None:
                try:
                    sv.comm_init(idbuf[0])
                except ha.HqError as e:
                    ok, errors[kind] = 0, str(e)
            else:
                ok = 0
            flush_c_stdio()  # RCCL prints a version banner through C stdio: out now, not after the JSON line
        elif kind == "ipc":
            try:
                mine = torch.frombuffer(bytearray(sv.comm_ipc_export()), dtype=torch.uint8)
            except ha.HqError as e:
                ok, errors[kind], mine = 0, str(e), torch.zeros(ha.capi.IPC_BLOB_BYTES, dtype=torch.uint8)
            every = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)                  # MPI_Allgather on comm_solver in the reference's world
            if everywhere(ok):
                try:
                    sv.comm_init_ipc([t.numpy().tobytes() for t in every])
                except ha.HqError as e:
                    ok, errors[kind] = 0, str(e)
            else:
                ok = 0
        else:
            sv.comm_init_host(gloo_exchange)
        up = everywhere(ok)
        if ok and not up:
            errors.setdefault(kind, "did not come up on every rank")
        return up

    def trial(sv, n=10):
        """ms per step of n steps (max over ranks), or None if the run failed anywhere."""
        ok = 1
        try:
            sv.run(3)
            sv.sync()
            dist.barrier()
            t0_ = time.perf_counter()
            sv.run(n)
            sv.sync()
            dist.barrier()
            ms = (time.perf_counter() - t0_) / n * 1e3
        except ha.HqError as e:
            ok, ms = 0, 0.0
            errors["trial"] = str(e)
        good = everywhere(ok)
        t_ms = torch.tensor([ms], dtype=torch.float64)
        dist.all_reduce(t_ms, op=dist.ReduceOp.MAX)
        return float(t_ms[0]) if good else None

    if world > 1:
        names = {"rccl": "RCCL grouped send/recv", "ipc": "IPC peer stores + epoch flags (hq_comm_init_ipc)",
                 "host": "host-staged (pinned buffers + gloo)"}
        def agree(a, b):
            """Same start, same trial steps, two transports: the fields must agree to the parity bar on every rank."""
            n_here = box.info["nharbored"]
            sample = np.unique(np.linspace(0, n_here - 1, min(n_here, 65536)).astype(np.int32))
            ga, gb = a.gather(sample)[0], b.gather(sample)[0]
            err = float(np.abs(ga - gb).max() / max(np.abs(ga).max(), 1e-300))
            return everywhere(err <= parity_tol(args))

        solver, chosen = select_transport(want, solver, new_solver, bring_up, trial, trials, agree)
        if trials:
            # the chosen context has run the trial's steps: back to the start field and step 0, so that the measured run
            # (its source window included) is the N = 1 run's (round-4 advisor)
            u1 = start_field()
            solver.upload(u1, u1 * (1.0 - 1e-3), 0)
            del u1
        rccl_ranks = int(solver.info()["nranks"]) if chosen == "rccl" else 0
        if chosen == "ipc":
            names["ipc"] += ", %s-grained receive arena" % ("coarse" if solver.info()["ipc_arena_coarse"] else "fine")
        transport = "%s, %d ranks" % (names[chosen], world) + \
                    "".join(" [%s failed: %s]" % (k, v) for k, v in errors.items())
        if rank == 0 and errors:
            print("bench.py: transport bring-up: %s" % errors, file=sys.stderr)
    repeats = max(1, args.repeats)
    total_steps = args.warmup + repeats * args.steps
    add_source(args, box, solver, octree, total_steps)
    info = solver.info()
    setup_s = time.perf_counter() - t_setup

    def barrier():
        if world > 1:
            dist.barrier()

    # Device pre-heat, not solver steps: a short run (the driver's W = 5 warm-up steps are 8 ms) would otherwise be timed on a
    # GPU that has not left its idle clocks.  solver_check_nan's kernel streams the displacement arrays and changes nothing.
    t_heat = time.perf_counter()
    while args.preheat > 0 and time.perf_counter() - t_heat < args.preheat:
        solver.check_finite()
    solver.run(args.warmup)
    solver.sync()
    # R back-to-back blocks of EXACTLY K steps, each bracketed by synchronize + barrier on both sides and reduced with
    # MAX over the ranks; the line reports the median block (a 20-step block of the 64M box is 20 ms: one block is
    # at the mercy of the box's clocks, round-5 review 7) and lists them all
    blocks = []
    for rep in range(repeats):
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        total_ms, kernel_ms = solver.run_timed(args.steps)     # enqueues K steps, HIP events, syncs
        torch.cuda.synchronize()
        barrier()
        blk = [time.perf_counter() - t0, kernel_ms, total_ms]
        if world > 1:
            t = torch.tensor(blk, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            blk = [float(v) for v in t]
        blocks.append(blk)
    elapsed, kernel_ms, total_ms = sorted(blocks)[(len(blocks) - 1) // 2]
    nonfinite = solver.check_finite()                      # solver_check_nan over the whole field
    info2 = solver.info()                                  # the phase split of the timed blocks
    if world > 1:
        t = torch.tensor([float(nonfinite)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        nonfinite = int(t[0])

    # parity that travels with the line: oracle cone windows on THIS context (partition + transport), behind the timed region
    parity = None
    if not args.no_parity:
        try:
            parity = parity_windows(args, box, solver, rank, world)
        except Exception as exc:                            # reported, never silently dropped
            parity = (0, 0, 1e300)
            print("bench.py: rank %d: parity windows failed: %s" % (rank, exc), file=sys.stderr)
        if world > 1:
            have = parity is not None
            t = torch.tensor([float(parity[0]) if have else 0.0, float(parity[1]) if have else 0.0], dtype=torch.float64)
            w = torch.tensor([parity[2] if have else 0.0], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            dist.all_reduce(w, op=dist.ReduceOp.MAX)
            if have:
                parity = (int(t[0]), int(t[1]), float(w[0]))
    parity_failed = parity is not None and not (parity[0] > 0 and parity[2] <= parity_tol(args))

    if rank == 0:
        E_total = box.info["total_elements"]
        E_local = box.info["lenum"]
        value = E_total * args.steps / elapsed
        is_patch = info["variant"] == ha.HQ_VARIANT_PATCH
        kernel = solver.dominant_kernel()
        # ONE basis for every workload (round-3 advisor finding): 72 B per node and step.  What a mesh with material of
        # its own in every element reads on top (24 B n_t row + 24 B coefficients) is reported beside it, not in `frac`.
        per_node = COMPULSORY_BYTES_PER_NODE * (0.5 if args.precision == "f32" else 1.0)      # a float state: 36 B
        compulsory = per_node * N                          # this rank's nodes: read u(t), u(t-dt), write u(t+dt)
        traffic = rd = wr = None
        source = None
        if pmc is not None and "error" not in pmc:
            got = traffic_of(pmc, kernel)
            if got:
                traffic, rd, wr, nprof = got
                source = ("rocprofv3 --pmc FETCH_SIZE (x2, gfx950) and --pmc WRITE_SIZE passes of this bench.py run, "
                          "%d steps each: %s and the patch kernels beside it" % (nprof, kernel))
        elif pmc is not None:
            source = "unmeasured: " + pmc["error"]
        elif world > 1:
            source = "unmeasured: counter passes run at N = 1 only"
        else:
            source = "unmeasured: --no-pmc"
        # frac: the bytes the step HAS to move (72 B per node: read u(t), u(t-dt), write u(t+dt)) over the time
        # it took, against the peak -- the fraction of the roofline the work needs.  counter_frac: the bytes the
        # kernels really moved (PMC), waste included.  wasted = the ratio of the two byte counts.
        # The time: HIP events on the compute stream around the step's kernels (kernel_ms) -- but a step whose shell runs
        # on a second stream beside the bricks (hq_options.brick_stream: the octree workloads) is longer than what one
        # stream's marks see, so the step's own device time (events around the whole block / K) bounds it from below:
        # frac <= 72 N / step time / peak always holds (round-5 review 4)
        step_ms = total_ms / args.steps
        roof_ms = max(kernel_ms, step_ms) if info2.get("brick_stream", 0) else kernel_ms
        achieved = compulsory / (roof_ms * 1e-3) / 1e9 if roof_ms > 0 else 0.0
        achieved_counter = traffic / (roof_ms * 1e-3) / 1e9 if (traffic is not None and roof_ms > 0) else None
        ideal_ms = compulsory / (HBM_PEAK_GBS * 1e9) * 1e3
        frac = achieved / HBM_PEAK_GBS
        counter_frac = achieved_counter / HBM_PEAK_GBS if achieved_counter is not None else None
        assert frac <= 1.0, "roofline fraction above 1: the byte count is not this kernel's traffic"
        assert counter_frac is None or counter_frac <= 1.0, "measured HBM rate above the peak: wrong counter arithmetic"
        out = {
            "metric": "element-updates/sec (whole node) + achieved HBM GB/s, 64M-elem box",
            "value": value,
            "unit": "element-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64" if args.precision == "f64" else "f32 state, f64 sums (the reference's -DSINGLE_PRECISION_SOLVER; libhq_solver_f32.so)",
            "data": "synthetic",
            "config": {"workload": WORKLOAD_NAMES[args.workload] + ("" if args.precision == "f64" else " -- SINGLE-PRECISION state (a separately named dtype, not the headline)"),
                       "elements": int(E_total),
                       "nodes": int(box.info["total_nodes"]), "partition": "octor block x%d" % world,
                       "kernel_variant": "patch" if is_patch else "scatter",
                       "patches": int(info["npatches"]), "stencil_patches": int(info["stencil_patches"]),
                       "ragged_patches": int(info["ragged_patches"]),
                       "patch_elements": int(info["patch_pairs"]),
                       "setup_s": round(setup_s, 1), "finite": nonfinite == 0, "rccl_ranks": rccl_ranks,
                       "transport": transport, "transport_trials_ms_per_step": trials or None,
                       "brick_nodes": int(info["brick_nodes"]), "brick_units_ragged": int(info.get("brick_units_ragged", 0)),
                       "preheat_s": args.preheat,
                       "repeats": repeats, "ms_per_step_runs": [round(b[0] / args.steps * 1e3, 5) for b in blocks],
                       # oracle cone windows stepped on this very context behind the timed region (parity_windows)
                       "parity_windows": parity[0] if parity else None,
                       "parity_nodes": parity[1] if parity else None,
                       "parity_worst": parity[2] if parity else None,
                       "parity_steps": (2 if octree else PARITY_STEPS) if parity else None, "parity_tol": parity_tol(args)},
            # achieved = HBM bytes the kernel really moved per launch (PMC, this session) / its mean launch
            # time (HIP events on its stream); where the counters are unavailable, the compulsory bytes
            # (a lower bound).  The reference formulation's 336 B per element-update is kept only as
            # algorithmic_equiv_GBs: the fused kernel never moves the force vector.
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": frac, "traffic": traffic, "traffic_read_corrected": rd, "traffic_write": wr,
                         "traffic_source": source,
                         "achieved_basis": "compulsory bytes: %d B per node and step" % int(per_node),
                         "counter_frac": counter_frac, "achieved_counter": achieved_counter,
                         "wasted": (traffic / compulsory) if traffic is not None else None,
                         "frac_incl_tables": (COMPULSORY_BYTES_PER_NODE_LATERAL / COMPULSORY_BYTES_PER_NODE * frac)
                                             if (args.workload in LATERAL or "gradient" in OCT_BASIN.get(args.workload, {})) else None,
                         "kernel": kernel, "kernel_ms": kernel_ms, "step_ms_events": step_ms, "roofline_ms": roof_ms,
                         # hq_info.t_*_us: the device-side split of a step (HIP events; the phases overlap)
                         "phase_us": {k: round(float(info2.get(k, 0.0)), 2) for k in
                                      ("t_step_us", "t_shell_us", "t_interior_us", "t_chain_us", "t_chain_exposed_us")},
                         "compulsory_bytes_per_launch": compulsory, "ideal_ms": ideal_ms,
                         "algorithmic_equiv_GBs": BYTES_PER_ELEMENT_UPDATE * value / 1e9,
                         "limiter": "see DESIGN.md s7"},
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
    solver.close()
    box.close()
    flush_c_stdio()
    barrier()                # every rank's library chatter is out before the one JSON line
    if rank == 0:
        print(json.dumps(out), flush=True)
        if parity_failed:
            print("bench.py: PARITY FAILED: %s windows, worst relative error %s (bar %g)" % (parity[0], parity[2], parity_tol(args)),
                  file=sys.stderr)
    if world > 1:
        dist.destroy_process_group()
    return 1 if parity_failed else 0


if __name__ == "__main__":
    sys.exit(main() or 0)
