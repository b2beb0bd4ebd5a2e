"""The oracle (oracle/herc_oracle.c) against the reference's golden vectors.

Pins the CPU restatement before anything is checked against it:
 * known-answer constants of SURVEY.md s8a (aBase/bBase, float-evaluated
   mu/lambda/M, K1/K2 entries),
 * full-field checkpoints written by the real reference binary for the
   effective and conventional stiffness methods and Rayleigh / mass / no damping,
 * the station traces the reference ships in examples/simple/expected-out.
"""
import numpy as np
import pytest

from oracle import herc_oracle as ho
from tests import helpers as H


def test_setab_known_answers():
    a, b = ho.setab(5.0, ho.DAMP_RAYLEIGH)
    assert a == 13.639003892573502
    assert b == 0.05897899877679408
    assert ho.setab(5.0, ho.DAMP_NONE) == (0.0, 0.0)
    am, bm = ho.setab(5.0, ho.DAMP_MASS)
    assert bm == 0.0 and am > 0


def test_K_known_answers():
    K1, K2 = ho.compute_K()
    assert np.allclose(np.diag(K1[0, 0]), 4.0, rtol=0, atol=1e-14)
    assert np.allclose(K1[0, 0][~np.eye(3, dtype=bool)], 0.75, rtol=0, atol=1e-14)
    assert np.allclose(np.diag(K2[0, 0]), 1.0, rtol=0, atol=1e-14)
    assert np.allclose(K2[0, 0][~np.eye(3, dtype=bool)], 0.75, rtol=0, atol=1e-14)
    assert np.allclose(np.diag(K1[0, 7]), -1.0, rtol=0, atol=1e-14)
    assert np.allclose(K1[0, 7][~np.eye(3, dtype=bool)], -0.375, rtol=0, atol=1e-14)
    # c1*K1 + c2*K2 is symmetric as a 24x24 matrix
    A = (1.3 * K1 + 0.7 * K2).transpose(0, 2, 1, 3).reshape(24, 24)
    assert np.abs(A - A.T).max() < 1e-14


def test_solver_init_float_evaluation():
    p = H.c1_problem()
    e = p["etable"][100]
    dt2 = 1e-3 * 1e-3
    # mu and lambda are evaluated in single precision first (psolve.c:3242-3248)
    assert e[0] == dt2 * 62.5 * 32398098432.0 / 9
    assert e[1] == dt2 * 62.5 * 32403800064.0 / 9
    zeta = 0.0028868359513580799
    b = zeta * 0.05897899877679408
    assert e[2] == b * 1e-3 * 62.5 * 32398098432.0 / 9
    # an interior node collects 8 x M = 8 x 82397464
    n_int = int(np.argmax(p["ntable"][:, 0]))
    assert p["ntable"][n_int, 0] == 8 * 82397464.0


@pytest.mark.parametrize("case,stiff,damping", [
    ("c1_short", ho.STIFF_EFFECTIVE, "rayleigh"),
    ("c1_conv", ho.STIFF_CONVENTIONAL, "rayleigh"),
    ("c1_none", ho.STIFF_EFFECTIVE, "none"),
    ("c1_mass", ho.STIFF_EFFECTIVE, "mass"),
])
def test_checkpoints_bitwise(case, stiff, damping):
    """Same loop order as the reference => identical bits, full field."""
    g = H.load(case)
    p = H.c1_problem(damping)
    tm1 = np.zeros((p["N"], 3))
    tm2 = np.zeros((p["N"], 3))
    ids, phi = ho.station_weights(H.C1_STATIONS, H.C1_H, H.C1_NX, H.C1_NY, H.C1_NZ,
                                  p["lnid"], p["elem_ijk"])
    done = 0
    caps = []
    for k, step in enumerate(g["ckpt_steps"]):
        caps.append(ho.solver_run(p["lnid"], p["etable"], p["ntable"], tm1, tm2, done,
                                  int(step) - done, p["dt"], damping=p["damping"],
                                  stiff_method=stiff, loaded_lnid=g["loaded_lnid"],
                                  forces=g["forces"], cap_lnid=ids))
        done = int(step)
        # checkpoint = state after the swap at the top of `step` (io_checkpoint.c:98-112)
        assert np.array_equal(tm1, g["ckpt_tm2"][k])
        assert np.array_equal(tm2, g["ckpt_tm1"][k])
    assert np.abs(g["ckpt_tm1"][-1]).max() > 100.0      # the wave is really there
    # station traces as the reference printed them ("% 8e": 7 significant digits)
    cap = np.concatenate(caps).reshape(done, len(H.C1_STATIONS), 8, 3)
    st = np.einsum("sn,tsnd->std", phi, cap)
    ref = g["stations"][:, :done, 1:]
    assert np.abs(st - ref).max() <= 6e-7 * np.abs(ref).max()


def test_fused_formulation_matches_reference_loops():
    """Formulation B (one effective product on u1 + beta (u1-u2)) is the same
    algebra as stiffness + Rayleigh damping loops: agreement ~1e-13."""
    g = H.load("c1_short")
    p = H.c1_problem()
    a1, a2 = np.zeros((p["N"], 3)), np.zeros((p["N"], 3))
    ho.solver_run(p["lnid"], p["etable"], p["ntable"], a1, a2, 0, 800, p["dt"], formulation=1,
                  loaded_lnid=g["loaded_lnid"], forces=g["forces"])
    assert H.rel_linf(a2, g["ckpt_tm1"][1]) < 1e-11
    assert H.rel_linf(a1, g["ckpt_tm2"][1]) < 1e-11


def test_zero_skip_does_not_change_results():
    g = H.load("c1_short")
    p = H.c1_problem()
    a1, a2 = np.zeros((p["N"], 3)), np.zeros((p["N"], 3))
    ho.solver_run(p["lnid"], p["etable"], p["ntable"], a1, a2, 0, 400, p["dt"], zero_skip=False,
                  loaded_lnid=g["loaded_lnid"], forces=g["forces"])
    assert H.rel_linf(a2, g["ckpt_tm1"][0]) < 1e-13


def test_uniform_mesh_generator_is_octor_order():
    lnid, node_ijk, elem_ijk, _ = H.c1_mesh()
    e2, l2, n2 = ho.uniform_mesh(H.C1_NX, H.C1_NY, H.C1_NZ)
    assert np.array_equal(e2, elem_ijk) and np.array_equal(l2, lnid) and np.array_equal(n2, node_ijk)
    assert len(node_ijk) == 2601 and len(lnid) == 2048


def test_K_matches_reference_dump():
    """print_matrix_k = yes output of the reference (psolve.c:3183-3225), 3 digits."""
    import os
    import re
    txt = open(os.path.join(H.GOLDEN, "c1_full_stdout_K.txt")).read()
    K1, K2 = ho.compute_K()
    for name, K in (("K1", K1), ("K2", K2)):
        blk = txt[txt.index("Stiffness Matrix " + name):]
        rows = [l for l in blk.splitlines()[1:] if l.strip()][:24]
        ref = np.array([[float(v) for v in re.findall(r"[-+]?\d\.\d+e[-+]\d+", r)] for r in rows])
        assert ref.shape == (24, 24)
        mine = K.transpose(0, 2, 1, 3).reshape(24, 24)
        assert np.abs(mine - ref).max() < 5.1e-3 * np.abs(ref).max()


def test_full_run_checkpoints_and_shipped_station_traces():
    """20 000 steps of examples/simple: bit-identical to the real reference's
    checkpoints at steps 12 000 / 18 000, and equal to the station traces the
    reference SHIPS (examples/simple/expected-out/stations) to their printed
    precision."""
    g = H.load("c1_full")
    p = H.c1_problem()
    tm1, tm2 = np.zeros((p["N"], 3)), np.zeros((p["N"], 3))
    ids, phi = ho.station_weights(H.C1_STATIONS, H.C1_H, H.C1_NX, H.C1_NY, H.C1_NZ,
                                  p["lnid"], p["elem_ijk"])
    done, caps = 0, []
    for k, step in enumerate(list(g["ckpt_steps"]) + [20000]):
        caps.append(ho.solver_run(p["lnid"], p["etable"], p["ntable"], tm1, tm2, done,
                                  int(step) - done, p["dt"], loaded_lnid=g["loaded_lnid"],
                                  forces=g["forces"], cap_lnid=ids))
        done = int(step)
        if k < len(g["ckpt_steps"]):
            assert np.array_equal(tm1, g["ckpt_tm2"][k])
            assert np.array_equal(tm2, g["ckpt_tm1"][k])
    cap = np.concatenate(caps).reshape(20000, 5, 8, 3)
    st = np.einsum("sn,tsnd->std", phi, cap)
    exp = g["expected_every20"][:, :, 1:]
    mine = st[:, ::20, :][:, :exp.shape[1], :]
    scale = np.abs(exp).max()
    assert scale > 1000.0
    assert np.abs(mine - exp).max() <= 6e-7 * scale
    head = g["expected_head"][:, :, 1:]
    assert np.abs(st[:, :head.shape[1], :] - head).max() <= 6e-7 * np.abs(head).max()
    # and the real reference binary reproduced its own shipped traces when the fixture was made
    assert float(g["max_abs_run_vs_expected"]) < 1e-3


def test_two_level_mesh_with_hanging_nodes_bitwise():
    """SURVEY s8 a13 / config 5 in miniature: the REAL reference meshed a soft layer one
    level deeper (5632 elements, 7179 nodes, 800 dangling nodes).  Mixed-level element
    order, node order, node_setproperty classification, anchor lists, the mass
    distribution at init and compute_adjust every step: bit-identical checkpoints."""
    p = H.c5_problem()
    g = p["golden"]
    assert p["E"] == int(g["total_elements"]) == 5632
    assert p["N"] == int(g["total_nodes"]) == 7179
    ids, ptr, anchors = p["dangling"]
    assert len(ids) == int(g["total_dangling"]) == 800
    deps = np.diff(ptr)
    assert set(deps.tolist()) == {2, 4} and (deps == 4).sum() == 16 * 16
    tm1, tm2 = np.zeros((p["N"], 3)), np.zeros((p["N"], 3))
    done = 0
    for k, step in enumerate(g["ckpt_steps"]):
        ho.solver_run(p["lnid"], p["etable"], p["ntable"], tm1, tm2, done, int(step) - done, p["dt"],
                      loaded_lnid=g["loaded_lnid"], forces=g["forces"], dangling=p["dangling"])
        done = int(step)
        assert np.array_equal(tm1, g["ckpt_tm2"][k])
        assert np.array_equal(tm2, g["ckpt_tm1"][k])
    # a hanging node sits at the mean of its anchors after every step (psolve.c:5992-6035)
    k = 17
    assert np.allclose(tm2[ids[k]], tm2[anchors[ptr[k]:ptr[k + 1]]].mean(axis=0), rtol=1e-13, atol=0)


def test_two_level_generator_reproduces_the_reference_mesh():
    """tests/helpers.two_level_mesh (used for larger hanging-node cases) rebuilds exactly
    the mesh, tables and constants the reference produced for c5_two_level."""
    ref = H.c5_problem()
    mine = H.two_level_mesh(32, 32, 4, 6)
    assert np.array_equal(mine["lnid"], ref["lnid"])
    assert np.array_equal(mine["etable"], ref["etable"]) and np.array_equal(mine["ntable"], ref["ntable"])
    for a, b in zip(mine["dangling"], ref["dangling"]):
        assert np.array_equal(a, b)


def test_three_level_mesh_all_material_branches_bitwise():
    """The reference on a three-material model that takes every branch of mu_and_lambda
    (Vp/Vs cap; negative lambda -> Vp rewritten) and the damping threshold, meshed by its own
    Vs rule on THREE octree levels (592 elements, 264 hanging nodes): bit-identical."""
    p = H.c5_problem("c5_three_level")
    g = p["golden"]
    assert p["E"] == int(g["total_elements"]) == 592 and p["N"] == int(g["total_nodes"]) == 973
    assert len(p["dangling"][0]) == int(g["total_dangling"]) == 264
    assert sorted(set(p["elem_size"].tolist())) == [1, 2, 4]
    mats = {tuple(m) for m in g["mat_vs_vp_rho"].tolist()}
    assert (150.0, 1500.0, 1800.0) in mats and (2000.0, 2500.0, 2300.0) in mats
    tm1, tm2 = np.zeros((p["N"], 3)), np.zeros((p["N"], 3))
    done = 0
    for k, step in enumerate(g["ckpt_steps"]):
        ho.solver_run(p["lnid"], p["etable"], p["ntable"], tm1, tm2, done, int(step) - done, p["dt"],
                      loaded_lnid=g["loaded_lnid"], forces=g["forces"], dangling=p["dangling"])
        done = int(step)
        assert np.array_equal(tm1, g["ckpt_tm2"][k]) and np.array_equal(tm2, g["ckpt_tm1"][k])
    assert np.abs(tm2).max() > 10.0


def test_layered_model_mesh_checkpoints_bitwise():
    """The reference on a three-material layered model whose mesh needs 2:1 balancing (edges of
    31.25 / 62.5 / 125 m, 1008 hanging nodes, two materials inside one level): bit-identical."""
    p = H.c5_problem("c5_layered")
    g = p["golden"]
    assert p["E"] == int(g["total_elements"]) == 2944 and len(p["dangling"][0]) == int(g["total_dangling"]) == 1008
    tm1, tm2 = np.zeros((p["N"], 3)), np.zeros((p["N"], 3))
    done = 0
    for k, step in enumerate(g["ckpt_steps"]):
        ho.solver_run(p["lnid"], p["etable"], p["ntable"], tm1, tm2, done, int(step) - done, p["dt"],
                      loaded_lnid=g["loaded_lnid"], forces=g["forces"], dangling=p["dangling"])
        done = int(step)
        assert np.array_equal(tm1, g["ckpt_tm2"][k]) and np.array_equal(tm2, g["ckpt_tm1"][k])
    assert np.abs(tm2).max() > 1.0


def test_basin_mesh_lateral_refinement_bitwise():
    """BASELINE config 5's mesh class: the reference on a LATERALLY varying model (a dipping sediment
    wedge + a soft box against a domain face; tests/golden/make_golden.py case_basin) -- its Vs rule
    and 2:1 balancing leave refinement interfaces with x-, y- and z-normal faces and staircase
    corners.  Hanging nodes of EVERY orientation (mid-edge on x / y / z edges, mid-face on xy / xz / yz
    faces, also on domain faces) are classified as node_setproperty does (octor.c:3294) and the
    checkpoints are bit-identical."""
    p = H.c5_problem("c5_basin")
    g = p["golden"]
    assert p["E"] == int(g["total_elements"]) == 5429 and p["N"] == int(g["total_nodes"]) == 7046
    ids, ptr, anchors = p["dangling"]
    assert len(ids) == int(g["total_dangling"]) == 1196
    assert sorted(set(p["elem_size"].tolist())) == [1, 2, 4]
    # the orientation of every hanging node from its anchors: edge nodes along x / y / z, face nodes normal to x / y / z
    q = p["node_q"].astype(np.int64)
    kinds = set()
    for k in range(len(ids)):
        a = q[anchors[ptr[k]:ptr[k + 1]]]
        varies = tuple(int(a[:, d].min() != a[:, d].max()) for d in range(3))
        kinds.add((len(a), varies))
    assert kinds == {(2, (1, 0, 0)), (2, (0, 1, 0)), (2, (0, 0, 1)), (4, (1, 1, 0)), (4, (1, 0, 1)), (4, (0, 1, 1))}
    far = q.max(axis=0)
    on_face = ((q[ids] == 0) | (q[ids] == far)).any(axis=1)
    assert on_face.sum() > 20                               # hanging nodes ON the domain's faces
    tm1, tm2 = np.zeros((p["N"], 3)), np.zeros((p["N"], 3))
    done = 0
    for k, step in enumerate(g["ckpt_steps"]):
        ho.solver_run(p["lnid"], p["etable"], p["ntable"], tm1, tm2, done, int(step) - done, p["dt"],
                      loaded_lnid=g["loaded_lnid"], forces=g["forces"], dangling=p["dangling"])
        done = int(step)
        assert np.array_equal(tm1, g["ckpt_tm2"][k]) and np.array_equal(tm2, g["ckpt_tm1"][k])
    assert np.abs(tm2).max() > 100.0


def test_gradient_basin_every_octant_its_own_material_bitwise():
    """The same basin with a velocity gradient (make_cvm `grad`; tests/golden/make_golden.py GRADIENT_CVM): every database
    octant has a material of its own, as any real CVM gives (setrec's 27-sample average, psolve.c:1307-1397) -- 286
    distinct (Vp, Vs, rho) over the 5 429 elements, no two neighbouring coarse elements alike -- so solver_init's
    single-precision material arithmetic (mu_and_lambda psolve.c:3236-3272, zeta = 10 / Vs) is exercised on hundreds of
    values instead of three.  The reference's checkpoints bit for bit."""
    p = H.c5_problem("c5_gradient")
    g = p["golden"]
    assert p["E"] == int(g["total_elements"]) == 5429 and len(p["dangling"][0]) == int(g["total_dangling"]) == 1196
    assert len(np.unique(g["mat_vs_vp_rho"], axis=0)) > 250 and len(np.unique(p["etable"], axis=0)) > 250
    tm1, tm2 = np.zeros((p["N"], 3)), np.zeros((p["N"], 3))
    done = 0
    for k, step in enumerate(g["ckpt_steps"]):
        ho.solver_run(p["lnid"], p["etable"], p["ntable"], tm1, tm2, done, int(step) - done, p["dt"],
                      loaded_lnid=g["loaded_lnid"], forces=g["forces"], dangling=p["dangling"])
        done = int(step)
        assert np.array_equal(tm1, g["ckpt_tm2"][k]) and np.array_equal(tm2, g["ckpt_tm1"][k])
    assert np.abs(tm2).max() > 100.0


@pytest.mark.parametrize("name", ["c5_basin_np8", "c5_basin_np5", "c5_gradient_np8"])
def test_basin_mesh_on_several_ranks_tables_and_fields(name):
    """The basin mesh on 8 and on 5 MPI ranks of the reference: block partition of a mixed-level leaf
    list, ownership, direct and indirect sharing across x / y / z-normal level interfaces, hanging nodes
    whose anchors live on other ranks -- octor's per-rank statistics and psolve's schedule summary
    EXACTLY, per-rank checkpoint stripes BIT FOR BIT (round 6: with the messenger lists in schedule_build's own order --
    a new messenger at the head of its list, a vertex's sharers in the order its owner met them as neighbours -- every sum
    of every exchange runs in the reference's order; before that: 1e-12)."""
    import re
    pr = H.c5_np8_problem(name)
    g, parts, nr = pr["golden"], pr["parts"], pr["nranks"]
    rows = [[int(v) for v in l.split()] for l in str(g["stat_mesh"]).splitlines() if re.match(r"^\d{6}\s", l)]
    sch = [[int(v) for v in l.split()] for l in str(g["stat_sched"]).splitlines() if re.match(r"^\s+\d+\s+\d+\s+\d+", l)]
    assert len(rows) == nr and len(sch) == nr
    for p, row, sc in zip(parts, rows, sch):
        r = p["rank"]
        assert row == [r, len(p["elems"]), int((p["owner"] == r).sum()), len(p["dangling"][0]), len(p["nodes"])]
        cnt = lambda lst: (len(lst), sum(len(v) for _, v in lst))
        assert sc[:9] == [r, *cnt(p["dn_sched"]["c"]), *cnt(p["dn_sched"]["s"]), *cnt(p["an_sched"]["c"]), *cnt(p["an_sched"]["s"])]
        assert np.array_equal(g["elem_ticks_%d" % r], pr["base"]["elem_ticks"][p["elems"]])
    assert sum(r_[3] for r_ in rows) == 1196 and any(r_[2] for r_ in sch)     # hanging nodes shared between ranks
    tm1s = [np.zeros((len(p["nodes"]), 3)) for p in parts]
    tm2s = [np.zeros((len(p["nodes"]), 3)) for p in parts]
    done = 0
    for step in g["ckpt_steps"]:
        ho.multi_rank_run(parts, pr["ets"], pr["nts"], tm1s, tm2s, done, int(step) - done, pr["dt"],
                          pr["loaded"], pr["forces"])
        done = int(step)
        for p in parts:
            ref2, ref1 = H.np8_stripe(g, step, p["rank"], len(p["nodes"]))
            assert np.array_equal(tm1s[p["rank"]], ref2) and np.array_equal(tm2s[p["rank"]], ref1)
    assert max(np.abs(t).max() for t in tm2s) > 100.0


def test_octree_mesh_on_eight_ranks_tables_and_fields():
    """The reference ran its two-level mesh on 8 MPI ranks.  octor's multi-rank tables
    restated from the global view (block partition, ownership by containing leaf, direct +
    indirect sharing, dnodeTable of owned hanging nodes, an/dn schedules) reproduce its
    per-rank statistics EXACTLY, and the multi-rank oracle (mass exchange at init, the four
    exchanges + compute_adjust per step) its per-rank checkpoint stripes BIT FOR BIT."""
    import re
    pr = H.c5_np8_problem()
    g, parts = pr["golden"], pr["parts"]
    rows = [[int(v) for v in l.split()] for l in str(g["stat_mesh"]).splitlines() if re.match(r"^\d{6}\s", l)]
    sch = [[int(v) for v in l.split()] for l in str(g["stat_sched"]).splitlines() if re.match(r"^\s+\d+\s+\d+\s+\d+", l)]
    assert len(rows) == 8 and len(sch) == 8
    for p, row, sc in zip(parts, rows, sch):
        r = p["rank"]
        assert row == [r, len(p["elems"]), int((p["owner"] == r).sum()), len(p["dangling"][0]), len(p["nodes"])]
        cnt = lambda lst: (len(lst), sum(len(v) for _, v in lst))
        mine = [r, *cnt(p["dn_sched"]["c"]), *cnt(p["dn_sched"]["s"]), *cnt(p["an_sched"]["c"]), *cnt(p["an_sched"]["s"])]
        assert sc[:9] == mine
        assert np.array_equal(g["elem_ticks_%d" % r], pr["base"]["elem_ticks"][p["elems"]])
    assert int(g["nharboredmax"]) == max(len(p["nodes"]) for p in parts)
    tm1s = [np.zeros((len(p["nodes"]), 3)) for p in parts]
    tm2s = [np.zeros((len(p["nodes"]), 3)) for p in parts]
    done = 0
    for step in g["ckpt_steps"]:
        ho.multi_rank_run(parts, pr["ets"], pr["nts"], tm1s, tm2s, done, int(step) - done, pr["dt"],
                          pr["loaded"], pr["forces"])
        done = int(step)
        for p in parts:
            ref2, ref1 = H.np8_stripe(g, step, p["rank"], len(p["nodes"]))
            assert np.array_equal(tm1s[p["rank"]], ref2) and np.array_equal(tm2s[p["rank"]], ref1)


def test_uniform_box_on_eight_ranks_bit_for_bit():
    """examples/simple on 8 MPI ranks of the reference (tests/golden/c1_np8: the domain's centre node is shared by all eight,
    its edges by four): the multi-rank oracle -- every rank's solver_init, the mass exchange, two exchanges per step in the
    messengers' list order -- against the per-rank checkpoints at steps 400 and 800, BIT FOR BIT.  This is the case that pins
    the ORDER of the lists: a vertex's share list holds the ranks in the order its owner met them as neighbours
    (com_allocpctl's scan, octor.c:2640-2741), schedule_build puts a new messenger at the head of its list, and
    schedule_senddata adds what arrives messenger by messenger; ascending or descending rank gives 1e-14, not zero."""
    g, base = H.load("c1_np8"), H.load("c1_short")
    m = ho.octree_mesh_from_elem_ticks(base["elem_ticks"], H.C1_FAR_TICKS)
    parts = ho.octree_partition(m, 8, [f // m["emin"] for f in H.C1_FAR_TICKS])
    mat = base["mat_vs_vp_rho"]
    edata = np.empty((len(m["lnid"]), 4), np.float32)
    edata[:, 0] = (1000.0 / 2 ** 30 * m["emin"] * m["elem_size"].astype(np.float64)).astype(np.float32)
    edata[:, 1], edata[:, 2], edata[:, 3] = mat[:, 1], mat[:, 0], mat[:, 2]
    eds = [np.ascontiguousarray(edata[p["elems"]]) for p in parts]
    fcs = [np.ascontiguousarray(m["face"][p["elems"]]) for p in parts]
    ets, nts = ho.multi_rank_init(parts, eds, fcs, 1e-3, 5.0)
    for r, p in enumerate(parts):
        assert np.array_equal(g["elem_ticks_%d" % r], base["elem_ticks"][p["elems"]])
    assert max(len(p["an_sched"]["s"]) for p in parts) == 7          # the owner of the centre node hears from all the others
    tm1s = [np.zeros((len(p["nodes"]), 3)) for p in parts]
    tm2s = [np.zeros((len(p["nodes"]), 3)) for p in parts]
    loaded, forces = [g["loaded_lnid_%d" % r] for r in range(8)], [g["forces_%d" % r] for r in range(8)]
    done = 0
    for step in g["ckpt_steps"]:
        ho.multi_rank_run(parts, ets, nts, tm1s, tm2s, done, int(step) - done, 1e-3, loaded, forces)
        done = int(step)
        for r, p in enumerate(parts):
            n = len(p["nodes"])
            assert np.array_equal(tm1s[r], g["ckpt%d_tm2_%d" % (done, r)][:n])
            assert np.array_equal(tm2s[r], g["ckpt%d_tm1_%d" % (done, r)][:n])
    assert max(np.abs(t).max() for t in tm2s) > 100.0


def test_output_planes_against_the_reference_files():
    """planedisplacements.<i> as the REAL reference wrote them (two planes, one dipping 60 degrees
    at strike 30, every 50 steps): plane geometry (compute_domain_coords_linearinterp,
    compute_global_coords), containing element + trilinear weights, and the field itself."""
    g = H.load("c1_planes")
    p = H.c1_problem()
    lonc, latc = g["surface_corners_lon_lat"][:, 0], g["surface_corners_lon_lat"][:, 1]
    planes = []
    for spec in g["plane_specs"]:
        lat, lon, depth, ds, ns, dd, nd, strike, dip = spec
        x, y = ho.domain_coords_linearinterp(lon, lat, lonc, latc, g["domain_xyz"][1], g["domain_xyz"][0])
        pts = ho.plane_points((x, y, depth), ds, int(ns), dd, int(nd), strike, dip)
        assert pts.min() >= 0 and (pts <= g["domain_xyz"]).all()
        planes.append(ho.station_weights(pts, H.C1_H, H.C1_NX, H.C1_NY, H.C1_NZ, p["lnid"], p["elem_ijk"]))
    x, y = ho.domain_coords_linearinterp(200.0, 300.0, lonc, latc, 1000.0, 1000.0)
    assert abs(x - 300.0) < 1e-9 and abs(y - 200.0) < 1e-9      # x <- latitude, y <- longitude
    nsteps = int(round(float(g["end_time"]) / float(g["dt"])))
    tm1 = np.zeros((p["N"], 3))
    tm2 = np.zeros((p["N"], 3))
    cap_ids = np.concatenate([ids.reshape(-1) for ids, _ in planes])
    cap = ho.solver_run(p["lnid"], p["etable"], p["ntable"], tm1, tm2, 0, nsteps, p["dt"],
                        damping=p["damping"], loaded_lnid=g["loaded_lnid"], forces=g["forces"],
                        cap_lnid=cap_ids)
    rate = int(g["plane_rate"])
    off = 0
    for i, (ids, phi) in enumerate(planes):
        ref = g["plane%d" % i]
        assert ref.shape[0] == (nsteps + rate - 1) // rate
        n = ids.size
        for k in range(ref.shape[0]):
            u = cap[k * rate, off:off + n].reshape(len(ids), 8, 3)
            got = np.zeros((len(ids), 3))
            for c in range(8):
                got += phi[:, c:c + 1] * u[:, c]
            assert np.abs(got - ref[k]).max() <= 1e-12 * max(np.abs(ref[k]).max(), 1e-300)
        off += n
        assert np.abs(ref[-1]).max() > 1e-3


def test_station_velocities_and_accelerations_as_the_reference_prints_them():
    """tests/golden/c1_stations_va: the reference run with print_station_velocities and
    print_station_accelerations = yes.  The oracle's loops give the station rows of tm1 at every
    step; tm2 / tm3 at a print are the rows one / two steps earlier (zero before the start: calloc,
    psolve.c:3347), and station_kinematics / station_line restate psolve.c:6705-6787.  The text is
    the reference's, line for line."""
    g = H.load("c1_stations_va")
    p = H.c1_problem()
    steps = g["stations"].shape[1]
    tm1 = np.zeros((p["N"], 3))
    tm2 = np.zeros((p["N"], 3))
    ids, phi = ho.station_weights(H.C1_STATIONS, H.C1_H, H.C1_NX, H.C1_NY, H.C1_NZ, p["lnid"], p["elem_ijk"])
    cap = ho.solver_run(p["lnid"], p["etable"], p["ntable"], tm1, tm2, 0, steps, p["dt"], loaded_lnid=g["loaded_lnid"],
                        forces=g["forces"], cap_lnid=ids).reshape(steps, len(ids), 8, 3)
    zero = np.zeros((8, 3))
    ref_lines = str(g["station0_text"]).split("\n")
    assert ref_lines[0] == ho.station_header(2)
    worst = 0.0
    # printed with 7 significant digits: compare in units of each quantity's largest value
    scale = np.repeat([np.abs(g["stations"][:, :, 1 + 3 * k:4 + 3 * k]).max() for k in range(3)], 3)
    for t in range(steps):
        for s in range(len(ids)):
            v = ho.station_kinematics(phi[s], cap[t, s], cap[t - 1, s] if t >= 1 else zero,
                                      cap[t - 2, s] if t >= 2 else zero, p["dt"], 2)
            worst = max(worst, float((np.abs(v - g["stations"][s, t, 1:]) / scale).max()))
            if s == 0 and t + 1 < len(ref_lines):
                assert ho.station_line(t * p["dt"], v) == "\n" + ref_lines[t + 1]
    assert worst <= 6e-7                                  # "% 8e": 7 significant digits
    assert np.abs(g["stations"][:, :, 7:]).max() > 1e3    # accelerations are really there
