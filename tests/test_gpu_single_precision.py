"""The reference's -DSINGLE_PRECISION_SOLVER (psolve.h:60-64: solver_float = float) as a separately named dtype:
libhq_solver_f32.so = the same sources with -DHQ_SINGLE_PRECISION_SOLVER -- the C-ABI takes and returns floats where the
reference's arrays are solver_float (tm1 / tm2, the n_t rows), the device keeps the state in floats (36 instead of 72
compulsory bytes per node and step), every sum inside a kernel stays double.

Checked against (i) checkpoints the REAL reference built with the switch wrote (oracle/_ref/psolve_f32, tests/golden/*_f32.npz)
and (ii) the oracle's own float build (oracle/libherc_oracle_f32.so, pinned on those checkpoints bit for bit in
tests/test_oracle_single_precision.py).

TOLERANCE, stated: the reference's float build rounds every local force, every n_t sum and the state; this path rounds the
state only.  On the 800-step run of examples/simple the reference's float build is 7.3e-6 (relative, L-inf) away from its own
double build; this path must stay within 2e-5 of the float reference there, and within 2e-6 of the float oracle over a few
steps from a seeded field.  Never the headline: the fp64 library is byte-for-byte what it was."""
import numpy as np
import pytest

import hercules_amd as ha
from hercules_amd import host
from oracle import herc_oracle as ho
from tests import helpers as H

pytestmark = pytest.mark.gpu

TOL_RUN = 2e-5       # hundreds of steps, against the reference's own float checkpoints
TOL_STEPS = 2e-6     # a few steps from a seeded field, against the float oracle on the same tables


def _ticks(node_ijk, edge=1 << 26):
    return (np.asarray(node_ijk, np.int64) * edge).astype(np.int32)


@pytest.mark.parametrize("variant", [ha.HQ_VARIANT_SCATTER, ha.HQ_VARIANT_PATCH])
def test_uniform_box_against_the_float_references_checkpoints(variant):
    g = H.load("c1_f32")
    p = H.c1_problem("rayleigh", real=np.float32)
    s = ha.Solver(p["lnid"], p["etable"], p["ntable"], p["dt"], node_xyz=_ticks(p["node_ijk"]), variant=variant, precision="f32")
    s.set_source(g["loaded_lnid"], g["forces"])
    done, worst = 0, 0.0
    for k, step in enumerate(g["ckpt_steps"]):
        s.run(int(step) - done)
        done = int(step)
        tm1, tm2 = s.download()
        assert tm1.dtype == np.float32 and tm2.dtype == np.float32
        worst = max(worst, H.rel_linf(tm1.astype(np.float64), g["ckpt_tm1"][k].astype(np.float64)),
                    H.rel_linf(tm2.astype(np.float64), g["ckpt_tm2"][k].astype(np.float64)))
    assert s.check_finite() == 0
    s.close()
    assert worst < TOL_RUN, worst
    # ... and closer to the DOUBLE reference than the float reference is (double sums, float state)
    g64 = H.load("c1_short")
    assert H.rel_linf(tm1.astype(np.float64), g64["ckpt_tm1"][-1]) < 2e-5


def test_two_level_octree_against_the_float_references_checkpoints():
    """compute_adjust on float tables: 800 hanging nodes, the reference's own float run of the two-level mesh."""
    p = H.c5_problem("c5_two_level_f32", real=np.float32)
    g = p["golden"]
    for variant in (ha.HQ_VARIANT_SCATTER, ha.HQ_VARIANT_PATCH):
        s = ha.Solver(p["lnid"], p["etable"], p["ntable"], p["dt"], dangling=p["dangling"],
                      node_xyz=(p["node_q"].astype(np.int64) * p["emin"]).astype(np.int32), variant=variant, precision="f32")
        s.set_source(g["loaded_lnid"], g["forces"])
        done = 0
        for k, step in enumerate(g["ckpt_steps"]):
            s.run(int(step) - done)
            done = int(step)
            tm1, tm2 = s.download()
            assert H.rel_linf(tm1.astype(np.float64), g["ckpt_tm1"][k].astype(np.float64)) < TOL_RUN
            assert H.rel_linf(tm2.astype(np.float64), g["ckpt_tm2"][k].astype(np.float64)) < TOL_RUN
        s.close()


def test_from_the_float_references_leaves_to_its_checkpoints_through_the_c_host():
    """The whole float path without the oracle's tables: the leaves of the mesh the float reference made (its element dump)
    -> hqh_mesh_from_leaves with solver_float = 4 (connectivity, hanging nodes, e_t, the n_t rows in the float build's own
    sums incl. the mass the hanging nodes hand to their anchors) -> libhq_solver_f32.so -> the float reference's checkpoints."""
    g = H.load("c5_two_level_f32")
    et = g["elem_ticks"]
    edge = et[:, 7, 0] - et[:, 0, 0]
    mat = g["mat_vs_vp_rho"]
    edata = np.empty((len(et), 4), np.float32)
    edata[:, 0] = (edge * (1000.0 / 2 ** 30)).astype(np.float32)
    edata[:, 1], edata[:, 2], edata[:, 3] = mat[:, 1], mat[:, 0], mat[:, 2]
    ob = host.OctBox.from_leaves(et[:, 0, :], edge, edata, H.C1_FAR_TICKS, 1e-3, float(g["freq"]), solver_float=4)
    p = H.c5_problem("c5_two_level_f32", real=np.float32)
    assert np.array_equal(ob.lnid, p["lnid"]) and np.array_equal(ob.ntable.astype(np.float32), p["ntable"])
    s = ob.create_solver(variant=ha.HQ_VARIANT_PATCH, precision="f32")
    s.set_source(g["loaded_lnid"], g["forces"])
    done = 0
    for k, step in enumerate(g["ckpt_steps"]):
        s.run(int(step) - done)
        done = int(step)
        tm1, tm2 = s.download()
        assert tm1.dtype == np.float32
        assert H.rel_linf(tm1.astype(np.float64), g["ckpt_tm1"][k].astype(np.float64)) < TOL_RUN
        assert H.rel_linf(tm2.astype(np.float64), g["ckpt_tm2"][k].astype(np.float64)) < TOL_RUN
    s.close()
    ob.close()


def _field32(box, seed, amp=1e-3):
    ijk = box.node_ijk.astype(np.int64)
    gid = (ijk[:, 2] * (box.ny + 1) + ijk[:, 1]) * (box.nx + 1) + ijk[:, 0]
    u = np.empty((len(gid), 3))
    for d in range(3):
        x = (gid * 3 + d + seed) * np.int64(2654435761) % np.int64(2 ** 31)
        u[:, d] = (x.astype(np.float64) / 2 ** 30 - 1.0) * amp
    return u.astype(np.float32)


def test_one_million_elements_bricks_against_the_float_oracle_and_in_eight_partitions():
    """The 1 M-element box: hq_k_brick on a float state (z faces with their columns), the shell's patches, gather /
    gather3 / upload in floats -- against the oracle's float build stepping the same tables: the C host side's with
    solver_float = 4, the float reference's own sums (tests/test_host_float_tables.py); then cut 8 ways (in-process
    transport: the records travel as doubles, what arrives is rounded as the owner's own copy is)."""
    from hercules_amd import capi
    nx, ny, nz, h, dt, freq = 128, 128, 64, 1000.0 / 128, 3.6e-4, 50.0
    box = host.Box(nx, ny, nz, h, dt, freq, solver_float=4)
    u = _field32(box, 4242)
    u2 = (0.999 * u.astype(np.float64)).astype(np.float32)
    nsteps = 3
    nt32 = np.ascontiguousarray(box.ntable, np.float32)
    assert np.array_equal(nt32.astype(np.float64), box.ntable)      # exact floats: nothing is rounded on the way over
    o1, o2 = u2.copy(), u.copy()                                    # oracle arrays are pre-swap
    ho.solver_run(box.lnid, box.etable.copy(), nt32, o1, o2, 0, nsteps, dt)
    s = box.create_solver(variant=ha.HQ_VARIANT_PATCH, tm1=u, tm2=u2, precision="f32")
    assert s.dominant_kernel() == "hq_k_brick" and s.info()["brick_nodes"] > 0.9 * len(u)
    s.run(nsteps)
    tm1, tm2 = s.download()
    assert tm1.dtype == np.float32
    assert H.rel_linf(tm1.astype(np.float64), o2.astype(np.float64)) < TOL_STEPS
    assert H.rel_linf(tm2.astype(np.float64), o1.astype(np.float64)) < TOL_STEPS
    ids = np.array([0, 17, len(u) // 2, len(u) - 1], np.int32)
    g1, g2, g3 = s.gather3(ids)
    assert g1.dtype == np.float32 and np.array_equal(g1, tm1[ids]) and np.array_equal(g2, tm2[ids])
    s.upload(u, u2, 0)
    s.run(nsteps)
    again, _ = s.download()
    assert np.array_equal(again, tm1)                                 # the same floats from the same start
    s.close()
    # the scatter kernels on the float state
    sc = box.create_solver(variant=ha.HQ_VARIANT_SCATTER, tm1=u, tm2=u2, precision="f32")
    sc.run(nsteps)
    s1, _ = sc.download()
    sc.close()
    assert H.rel_linf(s1.astype(np.float64), o2.astype(np.float64)) < TOL_STEPS
    gid = (box.node_ijk[:, 2].astype(np.int64) * (ny + 1) + box.node_ijk[:, 1]) * (nx + 1) + box.node_ijk[:, 0]
    lut = np.empty(gid.max() + 1, np.int64)
    lut[gid] = np.arange(len(gid))
    box.close()
    # cut 8 ways.  "The partitions give what the single context gives" needs the SAME rows on both sides, and with
    # solver_float = 4 a partition's rows are the N-rank float build's (other roundings than one rank's loop leaves,
    # tests/test_host_float_tables.py): this half runs on the double build's rows rounded to float, whole and cut
    box = host.Box(nx, ny, nz, h, dt, freq)
    nt32 = np.ascontiguousarray(box.ntable, np.float32)
    o1, o2 = u2.copy(), u.copy()
    ho.solver_run(box.lnid, box.etable.copy(), nt32, o1, o2, 0, nsteps, dt)
    s = box.create_solver(variant=ha.HQ_VARIANT_PATCH, tm1=u, tm2=u2, precision="f32")
    s.run(nsteps)
    tm1, _ = s.download()
    s.close()
    box.close()
    assert H.rel_linf(tm1.astype(np.float64), o2.astype(np.float64)) < TOL_STEPS
    parts = [host.Box(nx, ny, nz, h, dt, freq, rank=r, nranks=8) for r in range(8)]
    maps = [lut[(b.node_ijk[:, 2].astype(np.int64) * (ny + 1) + b.node_ijk[:, 1]) * (nx + 1) + b.node_ijk[:, 0]] for b in parts]
    solvers = [b.create_solver(tm1=u[m], tm2=u2[m], precision="f32") for b, m in zip(parts, maps)]
    capi.group_link(solvers)
    capi.group_run(solvers, nsteps)
    for sv, m in zip(solvers, maps):
        p1, p2 = sv.download()
        assert H.rel_linf(p1.astype(np.float64), o2[m].astype(np.float64)) < TOL_STEPS
        assert H.rel_linf(p1.astype(np.float64), tm1[m].astype(np.float64)) < 5e-7      # a float ulp or two from the single run
        sv.close()
    for b in parts:
        b.close()


@pytest.mark.parametrize("wl", ["o4s", "o4gs"])
def test_small_lateral_basin_float_state_against_the_float_oracle(wl):
    """o4s (laterally refined: full and ragged tile columns, element-form patches with hanging-node accumulators,
    hq_k_adjust_assign) on a float state, forces on 3 000 nodes, against the float oracle with compute_adjust.
    o4gs (round 6): the same basin with a velocity gradient -- the per-element kernels on a float state, full and RAGGED
    tiles (hq_k_brick_het<false, RAGGED> of libhq_solver_f32.so)."""
    import bench
    box, E, N, it = bench.make_octbox(wl, 0, 1)
    u = it["field"].astype(np.float32)
    u2 = (0.999 * it["field"]).astype(np.float32)
    nsteps = 3
    free = np.setdiff1d(np.arange(N, dtype=np.int64), box.dangling[0])
    loaded = free[np.linspace(0, len(free) - 1, 3000).astype(np.int64)].astype(np.int32)
    rng = np.random.default_rng(99)
    F = rng.uniform(-1.0, 1.0, (nsteps, len(loaded), 3)) * (1e-4 * np.abs(u).max() / box.dt ** 2) * box.ntable[loaded, 0][None, :, None]
    nt32 = np.ascontiguousarray(box.ntable, np.float32)
    o1, o2 = u2.copy(), u.copy()
    ho.solver_run(box.lnid, box.etable.copy(), nt32, o1, o2, 0, nsteps, box.dt, dangling=box.dangling, loaded_lnid=loaded, forces=F)
    for variant in (ha.HQ_VARIANT_PATCH, ha.HQ_VARIANT_SCATTER):
        s = box.create_solver(variant=variant, tm1=u, tm2=u2, precision="f32")
        if variant == ha.HQ_VARIANT_PATCH:
            assert s.info()["brick_units_ragged_het" if wl == "o4gs" else "brick_units_ragged"] > 0
            # (float n_t rows do not satisfy m2 = 2 m0 - (m0 - m1) to 1e-15: hq_create keeps such units unpacked, so this
            #  is the 24-byte form of the per-element kernels, hq_k_brick_het<false, RAGGED> among them)
            assert s.info()["brick_units_packed"] == 0
        s.set_source(loaded, F)
        s.run(nsteps)
        tm1, tm2 = s.download()
        s.close()
        assert H.rel_linf(tm1.astype(np.float64), o2.astype(np.float64)) < TOL_STEPS
        assert H.rel_linf(tm2.astype(np.float64), o1.astype(np.float64)) < TOL_STEPS
    box.close()


def test_the_two_libraries_keep_their_symbols_apart():
    """libhq_solver.so and libhq_solver_f32.so export the same names.  In a fresh process that loads the FLOAT build first,
    the C host side (libhq_host.so: hqh_solver_run calls hq_run / hq_gather / hq_download through its own references) must
    still drive the fp64 library -- the float build never joins the global symbol scope, and both are linked -Bsymbolic."""
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np
import hercules_amd as ha
from hercules_amd import capi, host
f32 = capi.load_library(precision="f32")                       # first in this process
assert f32.hq_real_bytes() == 4
box = host.Box(16, 16, 8, 62.5, 1e-3, 5.0)
loaded, pattern = box.point_source(500.0, 500.0, 100.0, 0.0, 90.0, 0.0)
rp = box.run_params(loaded=loaded, pattern=pattern, moment=1e15, rise_time=0.02, source_window=32)
F = box.source_table(rp, 0, 40)
a = box.create_solver()
box.solver_run(a, rp, 0, 40)                                   # libhq_host.so -> hq_* of the fp64 library
t1, t2 = a.download()
b = box.create_solver()
b.set_source(loaded, F)
b.run(40)                                                      # the same steps through ctypes
u1, u2 = b.download()
scale = np.abs(u1).max()
assert t1.dtype == np.float64 and scale > 0, (t1.dtype, scale)
assert np.isfinite(t1).all() and np.abs(t1 - u1).max() <= 1e-12 * scale and np.abs(t2 - u2).max() <= 1e-12 * scale, (np.abs(t1).max(), np.abs(t1 - u1).max(), scale)
c = box.create_solver(precision="f32")                         # and the float library still is the float library
c.set_source(loaded, F)
c.run(40)
w1, _ = c.download()
assert w1.dtype == np.float32 and np.abs(w1.astype(np.float64) - u1).max() < 1e-5 * np.abs(u1).max()
print("ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         universal_newlines=True, timeout=600, env=dict(os.environ, PYTHONPATH=root))
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-2000:]


@pytest.mark.parametrize("variant", [ha.HQ_VARIANT_SCATTER, ha.HQ_VARIANT_PATCH])
def test_eight_partitions_against_the_float_references_eight_rank_stripes(variant):
    """The float reference ran its two-level octree on 8 MPI ranks (tests/golden/c5_two_level_np8_f32: hanging nodes shared
    between ranks, float records in all four exchanges, the mass exchange on float n_t rows): eight float contexts in one
    process (in-process transport; the records travel as doubles and are rounded where the reference's land) against its
    per-rank checkpoint stripes, relative to the field's scale."""
    from hercules_amd import capi
    pr = H.c5_np8_problem("c5_two_level_np8_f32", real=np.float32)
    g, parts, mesh = pr["golden"], pr["parts"], pr["mesh"]
    solvers = []
    for p in parts:
        r = p["rank"]
        s = ha.Solver(p["lnid"], pr["ets"][r], pr["nts"][r], pr["dt"], dangling=p["dangling"], an_sched=p["an_sched"],
                      dn_sched=p["dn_sched"], rank=r, nranks=pr["nranks"], variant=variant,
                      node_xyz=mesh["node_q"][p["nodes"]], precision="f32")
        if len(pr["loaded"][r]):
            s.set_source(pr["loaded"][r], pr["forces"][r])
        solvers.append(s)
    capi.group_link(solvers)
    done = 0
    for step in g["ckpt_steps"]:
        capi.group_run(solvers, int(step) - done)
        done = int(step)
        stripes = [H.np8_stripe(g, step, p["rank"], len(p["nodes"])) for p in parts]
        scale = max(float(np.abs(ref1).max()) for _, ref1 in stripes)
        for (ref2, ref1), s in zip(stripes, solvers):
            tm1, tm2 = s.download()
            assert tm1.dtype == np.float32
            assert np.abs(tm1.astype(np.float64) - ref1).max() <= TOL_RUN * scale
            assert np.abs(tm2.astype(np.float64) - ref2).max() <= TOL_RUN * scale
    for s in solvers:
        s.close()


def test_eight_float_partitions_from_the_c_host_against_the_float_references_stripes():
    """The multi-rank float path without the oracle's tables: the leaves of the mesh -> hqh_mesh_from_leaves per rank (octor's
    partition, ownership, messenger lists; n_t rows with solver_float = 4: the N-rank float build's own sums -- every rank's
    elements apart, the three mass exchanges replayed, hq_host.c nt_rank_rows) -> eight contexts of libhq_solver_f32.so over
    the in-process transport -> the stripes the float reference's 8 MPI ranks wrote (its own per-rank force files).
    Measured (round 6): 2.3e-6 of the field's scale after 200 steps on these rows; 6e-5 on rows summed in ONE rank's order,
    1.8e-5 on the double build's rows rounded to float -- a float run feels which roundings its rows carry.  Bound: 5e-6."""
    from hercules_amd import capi
    g = H.load("c5_two_level_np8_f32")
    base = H.load(str(g["base"]))
    nranks = int(g["nranks"])
    et = base["elem_ticks"]
    edge = et[:, 7, 0] - et[:, 0, 0]
    mat = base["mat_vs_vp_rho"]
    edata = np.empty((len(et), 4), np.float32)
    edata[:, 0] = (edge * (1000.0 / 2 ** 30)).astype(np.float32)
    edata[:, 1], edata[:, 2], edata[:, 3] = mat[:, 1], mat[:, 0], mat[:, 2]
    boxes = [host.OctBox.from_leaves(et[:, 0, :], edge, edata, H.C1_FAR_TICKS, 1e-3, float(base["freq"]), rank=r, nranks=nranks,
                                     solver_float=4) for r in range(nranks)]
    solvers = [b.create_solver(precision="f32") for b in boxes]
    for r, s in enumerate(solvers):
        assert boxes[r].E == len(g["elem_ticks_%d" % r])               # octor's share of the elements
        if len(g["loaded_lnid_%d" % r]):
            s.set_source(g["loaded_lnid_%d" % r], g["forces_%d" % r])
    capi.group_link(solvers)
    done = 0
    for step in g["ckpt_steps"]:
        capi.group_run(solvers, int(step) - done)
        done = int(step)
        stripes = [H.np8_stripe(g, step, r, boxes[r].N) for r in range(nranks)]
        scale = max(float(np.abs(ref1).max()) for _, ref1 in stripes)
        for (ref2, ref1), s in zip(stripes, solvers):
            tm1, tm2 = s.download()
            assert tm1.dtype == np.float32
            assert np.abs(tm1.astype(np.float64) - ref1).max() <= 5e-6 * scale
            assert np.abs(tm2.astype(np.float64) - ref2).max() <= 5e-6 * scale
    for s in solvers:
        s.close()
    for b in boxes:
        b.close()
