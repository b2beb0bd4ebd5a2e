"""bench.py's roofline arithmetic on the counter CSVs committed under profiles/ (the two rocprofv3 --pmc passes
of a default bench run): HBM bytes per step = (FETCH_SIZE x 2 + WRITE_SIZE) KiB summed over the compute kernels
of a step, as MI355X_MICROARCH.md prescribes for gfx950, divided by the number of steps the counter pass RAN --
not by the dispatch count of a kernel the launcher may issue one to three times per step.  No GPU."""
import csv
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R02 = os.path.join(ROOT, "profiles", "r02")
R03 = os.path.join(ROOT, "profiles", "r03")


def _per_kernel(path, counter, drop=None):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        if drop and drop in r["Kernel_Name"]:
            continue
        a = out.setdefault(r["Kernel_Name"].split("(")[0], [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return out


@pytest.mark.parametrize("tag,line_file", [("default", "bench_default_v13.json"), ("v14", "bench_default_v14.json")])
def test_traffic_and_fraction_from_the_committed_counter_passes(tag, line_file):
    pmc = {"FETCH_SIZE": _per_kernel(os.path.join(R02, "pmc_%s_fetch_size.csv" % tag), "FETCH_SIZE"),
           "WRITE_SIZE": _per_kernel(os.path.join(R02, "pmc_%s_write_size.csv" % tag), "WRITE_SIZE")}
    line = json.load(open(os.path.join(R02, line_file)))
    kernel = line["roofline"]["kernel"]
    total, rd, wr, steps = bench.traffic_of(pmc, kernel)
    assert steps == bench.PMC_STEPS == 4 and rd > wr > 0
    # the committed line was computed from these very passes
    assert abs(total - line["roofline"]["traffic"]) <= 1e-6 * total
    nodes = line["config"]["nodes"]
    assert total >= bench.COMPULSORY_BYTES_PER_NODE * nodes * 0.99          # nothing can move less than the compulsory bytes
    assert abs(wr - 24.0 * nodes) <= 0.01 * wr                              # u(t+dt) is written exactly once: 24 B per node
    counter_frac = total / (line["roofline"]["kernel_ms"] * 1e-3) / 1e9 / bench.HBM_PEAK_GBS
    # round 2's lines called this `frac`; since round 3 it is `counter_frac` and `frac` is the compulsory-byte fraction
    assert 0.0 < counter_frac <= 1.0 and abs(counter_frac - line["roofline"]["frac"]) < 1e-9


@pytest.mark.parametrize("tag,line_file,kernel", [("v18", "bench_default_v18.json", "hq_k_brick"),
                                                  ("c3h_v19", "bench_c3h_v19.json", "hq_k_brick"),
                                                  ("v22", "bench_default_v22.json", "hq_k_brick"),
                                                  ("c3h_v22", "bench_c3h_v22.json", "hq_k_brick")])
def test_round3_lines_follow_from_their_counter_passes(tag, line_file, kernel):
    """Round 3's committed lines (bricks: hq_k_brick / hq_k_brick_het + the patch kernel of the faces beside it):
    `traffic`, `counter_frac`, `wasted` follow from the committed counter CSVs; `frac` is the compulsory-byte fraction."""
    pmc = {"FETCH_SIZE": _per_kernel(os.path.join(R03, "pmc_%s_fetch_size.csv" % tag), "FETCH_SIZE"),
           "WRITE_SIZE": _per_kernel(os.path.join(R03, "pmc_%s_write_size.csv" % tag), "WRITE_SIZE")}
    line = json.load(open(os.path.join(R03, line_file)))
    r = line["roofline"]
    assert r["kernel"] == kernel
    total, rd, wr, steps = bench.traffic_of(pmc, kernel)
    assert steps == 4 and abs(total - r["traffic"]) <= 1e-6 * total
    nodes = line["config"]["nodes"]
    comp = r["compulsory_bytes_per_launch"]
    assert comp in (bench.COMPULSORY_BYTES_PER_NODE * nodes, bench.COMPULSORY_BYTES_PER_NODE_LATERAL * nodes) and total >= 0.99 * comp
    assert abs(r["wasted"] - total / comp) < 1e-9
    t = r["kernel_ms"] * 1e-3
    assert abs(r["frac"] - comp / t / 1e9 / bench.HBM_PEAK_GBS) < 1e-9
    assert abs(r["counter_frac"] - total / t / 1e9 / bench.HBM_PEAK_GBS) < 1e-9
    assert 0.0 < r["frac"] <= r["counter_frac"] <= 1.0


def test_round6_line_follows_from_its_counter_passes():
    """The final tree's default line (profiles/r06/bench_default.json) and the two --pmc passes of that very run:
    traffic, counter_frac, wasted and frac recomputed; the written bytes are u(t + dt) once; the line carries the median of
    five timed blocks, its parity windows, the device-side phase split and the CPU baseline's element count."""
    R06 = os.path.join(ROOT, "profiles", "r06")
    pmc = {"FETCH_SIZE": _per_kernel(os.path.join(R06, "pmc_default_fetch_size.csv"), "FETCH_SIZE"),
           "WRITE_SIZE": _per_kernel(os.path.join(R06, "pmc_default_write_size.csv"), "WRITE_SIZE")}
    line = json.load(open(os.path.join(R06, "bench_default.json")))
    r, c = line["roofline"], line["config"]
    total, rd, wr, steps = bench.traffic_of(pmc, r["kernel"])
    assert steps == 4 and abs(total - r["traffic"]) <= 1e-6 * total
    nodes = c["nodes"]
    comp = bench.COMPULSORY_BYTES_PER_NODE * nodes
    assert r["compulsory_bytes_per_launch"] == comp and abs(wr - 24.0 * nodes) <= 0.01 * wr
    t = r["roofline_ms"] * 1e-3
    assert abs(r["frac"] - comp / t / 1e9 / bench.HBM_PEAK_GBS) < 1e-9 and abs(r["counter_frac"] - total / t / 1e9 / bench.HBM_PEAK_GBS) < 1e-9
    assert 0.5 < r["frac"] <= r["counter_frac"] <= 1.0 and abs(r["wasted"] - total / comp) < 1e-9
    runs = c["ms_per_step_runs"]
    assert c["repeats"] == len(runs) == 5 and abs(sorted(runs)[2] - line["ms_per_step"]) < 1e-4
    assert c["parity_windows"] >= 4 and c["parity_worst"] <= 1e-9
    assert r["phase_us"]["t_interior_us"] > 10 * r["phase_us"]["t_shell_us"] > 0
    assert line["cpu_baseline"]["kind"] == "reference" and line["cpu_baseline"]["elements"] == 8388608
    assert line["cpu_baseline"]["recorded_64m"]["elements"] == 67108864


def test_step_count_does_not_depend_on_which_kernels_a_step_launches():
    """hq_k_patch_stencil<512> runs twice per step and <768> once on the 64M box; a mesh without far-face cubes has
    no <768> rows at all.  The bytes per step must come out as the sum over the kernels divided by the steps the
    pass ran, whatever the launch mix."""
    f = os.path.join(R02, "pmc_v14_fetch_size.csv")
    w = os.path.join(R02, "pmc_v14_write_size.csv")
    full = {"FETCH_SIZE": _per_kernel(f, "FETCH_SIZE"), "WRITE_SIZE": _per_kernel(w, "WRITE_SIZE")}
    no768 = {"FETCH_SIZE": _per_kernel(f, "FETCH_SIZE", drop="768"), "WRITE_SIZE": _per_kernel(w, "WRITE_SIZE", drop="768")}
    assert any("768" in k for k in full["FETCH_SIZE"]) and not any("768" in k for k in no768["FETCH_SIZE"])
    t_full = bench.traffic_of(full, "hq_k_patch_stencil")
    t_no = bench.traffic_of(no768, "hq_k_patch_stencil")
    assert t_full[3] == t_no[3] == 4
    big = sum(t for k, (n, t) in full["FETCH_SIZE"].items() if "768" in k) * 1024.0 * 2.0 / 4 + \
        sum(t for k, (n, t) in full["WRITE_SIZE"].items() if "768" in k) * 1024.0 / 4
    assert big > 0 and abs((t_full[0] - t_no[0]) - big) <= 1e-9 * t_full[0]
    # a kernel name that never ran: no traffic figure rather than a wrong one
    assert bench.traffic_of(full, "hq_k_brick") is None
    # bricks + patch kernels of one step are summed
    synth = {c: {"hq_k_brick<false>": [4, 4000.0], "hq_k_patch_seed": [4, 400.0], "hq_k_pack": [8, 99.0]} for c in ("FETCH_SIZE", "WRITE_SIZE")}
    tot, rd, wr, steps = bench.traffic_of(synth, "hq_k_brick")
    assert steps == 4 and rd == 1100.0 * 1024 * 2 and wr == 1100.0 * 1024 and tot == rd + wr


# ---------------------------------------------------------------------------------------------
# bench.py --gpus N: which transport carries the halo records (control flow only; no GPU, no ranks)
# ---------------------------------------------------------------------------------------------

class _FakeSolver:
    made = 0

    def __init__(self):
        _FakeSolver.made += 1
        self.id, self.closed, self.transport = _FakeSolver.made, False, None

    def close(self):
        assert not self.closed
        self.closed = True


def _select(want, comes_up, ms, agree=None):
    """comes_up: {kind: bool}; ms: {kind: ms per step or None (the trial failed)}; agree: do the two trial contexts hold
    the same field (None: not asked)."""
    _FakeSolver.made = 0
    first = _FakeSolver()
    made = [first]

    def new_solver():
        made.append(_FakeSolver())
        return made[-1]

    def bring_up(kind, sv):
        assert not sv.closed and sv.transport is None          # never a second transport on a context, never a closed one
        if comes_up.get(kind, kind == "host"):
            sv.transport = kind
            return True
        sv.transport = "broken:" + kind
        return False

    trials = {}
    solver, chosen = bench.select_transport(want, first, new_solver, bring_up, lambda sv: ms[sv.transport], trials,
                                            None if agree is None else (lambda a, b: agree))
    alive = [s for s in made if not s.closed]
    assert alive == [solver] and solver.transport == chosen    # exactly the chosen context survives, and it carries it
    return chosen, trials


def test_transport_selection_of_a_multi_gpu_bench_run():
    # both device-side transports come up: both are timed, the faster one is kept
    assert _select("auto", {"rccl": True, "ipc": True}, {"rccl": 0.30, "ipc": 0.21}) == ("ipc", {"rccl": 0.30, "ipc": 0.21})
    assert _select("auto", {"rccl": True, "ipc": True}, {"rccl": 0.18, "ipc": 0.21})[0] == "rccl"
    # the faster IPC transport is REJECTED when its field differs from the RCCL context's after the same trial run
    chosen, tr = _select("auto", {"rccl": True, "ipc": True}, {"rccl": 0.30, "ipc": 0.21}, agree=False)
    assert chosen == "rccl" and "ipc_rejected" in tr
    assert _select("auto", {"rccl": True, "ipc": True}, {"rccl": 0.30, "ipc": 0.21}, agree=True)[0] == "ipc"
    # a trial that fails (timeouts, a device error) takes its transport out
    assert _select("auto", {"rccl": True, "ipc": True}, {"rccl": None, "ipc": 0.4})[0] == "ipc"
    assert _select("auto", {"rccl": True, "ipc": True}, {"rccl": None, "ipc": None})[0] == "host"
    # only one comes up (ranks that share a GPU: RCCL refuses duplicate devices): no trial
    assert _select("auto", {"rccl": False, "ipc": True}, {}) == ("ipc", {})
    assert _select("auto", {"rccl": True, "ipc": False}, {}) == ("rccl", {})
    assert _select("auto", {"rccl": False, "ipc": False}, {}) == ("host", {})
    # explicit choices and their fall-backs
    assert _select("rccl", {"rccl": True}, {})[0] == "rccl"
    assert _select("rccl", {"rccl": False, "ipc": True}, {})[0] == "ipc"
    assert _select("rccl", {"rccl": False, "ipc": False}, {})[0] == "host"
    assert _select("ipc", {"ipc": True}, {})[0] == "ipc"
    assert _select("ipc", {"ipc": False}, {})[0] == "host"
    assert _select("host", {}, {})[0] == "host"


@pytest.mark.parametrize("wl", ["c1", "m1h"])
def test_parity_windows_of_the_bench_line_against_a_whole_box_oracle(monkeypatch, wl):
    """bench.parity_windows (the oracle cone windows every bench line carries, config.parity_*) with a stand-in for the
    device context that steps the WHOLE box with the oracle: a window built as a 16^3 box of its own must then agree to
    rounding at every checked node -- in the interior, at brick-tile borders and where windows touch the domain's faces,
    edges and corners (dashpots, free surface) -- and must notice a context that computes something else."""
    import argparse
    import bench
    from hercules_amd import host
    from oracle import herc_oracle as ho
    # (m1h: material of its own in every element -- the windows are boxes with hqh_box_params.origin, so that their
    #  classes are the big box's)
    monkeypatch.setitem(bench.WORKLOADS, wl, (128, 32, 32, 62.5, 1e-3, 5.0))
    nx, ny, nz, h, dt, freq = bench.WORKLOADS[wl]
    ncls, amp = bench.LATERAL.get(wl, (0, 0.0))
    box = host.Box(nx, ny, nz, h, dt, freq, lateral_classes=ncls, lateral_amp=amp)

    class WholeBoxOracle:
        wrong = 0.0

        def set_source(self, ids, F):
            assert len(ids) == 0

        def upload(self, tm1, tm2, step):
            self.u1, self.u2 = tm1.copy(), tm2.copy()

        def run(self, k):
            o1, o2 = self.u2.copy(), self.u1.copy()
            ho.solver_run(box.lnid, box.etable.copy(), box.ntable.copy(), o1, o2, 0, k, dt)
            self.u1, self.u2 = o2, o1
            self.u1[1000] += self.wrong

        def sync(self):
            pass

        def gather(self, ids):
            return self.u1[ids], self.u2[ids]

    args = argparse.Namespace(workload=wl)
    s = WholeBoxOracle()
    nwin, nchecked, worst = bench.parity_windows(args, box, s, 0, 1)
    assert nwin >= 4 and nchecked > 4 * 11 ** 3 and worst < 1e-13
    s.wrong = 1e-6
    checked_1000 = bench.parity_windows(args, box, s, 0, 1)[2]
    assert checked_1000 < 1e-13 or checked_1000 > 1e-9                  # node 1000 is either outside every window or caught
    box.close()


def test_octree_parity_windows_of_the_bench_line_against_a_whole_mesh_oracle():
    """bench.parity_windows_octree (one rank of an octree workload: windows centred on hanging nodes, cut out with the true
    table rows) with a stand-in context that steps the WHOLE laterally refined basin o4s with the oracle."""
    import argparse
    import bench
    from oracle import herc_oracle as ho
    box, E, N, it = bench.make_octbox("o4s", 0, 1)
    box.node_ijk, box.start_interfaces = box.node_xyz, it

    class WholeMeshOracle:
        def set_source(self, ids, F):
            assert len(ids) == 0

        def upload(self, tm1, tm2, step):
            self.u1, self.u2 = tm1.copy(), tm2.copy()

        def run(self, k):
            o1, o2 = self.u2.copy(), self.u1.copy()
            ho.solver_run(box.lnid, box.etable.copy(), box.ntable.copy(), o1, o2, 0, k, box.dt, dangling=box.dangling)
            self.u1, self.u2 = o2, o1

        def sync(self):
            pass

        def gather(self, ids):
            return self.u1[ids], self.u2[ids]

    nwin, nchecked, worst = bench.parity_windows(argparse.Namespace(workload="o4s"), box, WholeMeshOracle(), 0, 1)
    assert nwin >= 4 and nchecked > 1000 and worst < 1e-12

    # A PARTITION of the mesh (round-5 review 2a: an N > 1 octree line carries parity too).  The root rank builds the whole
    # mesh, cuts windows around hanging nodes the partitions SHARE and hands (node key, expected value) records to every
    # rank; here one process plays the root over a communicator of one, its "context" is rank 1 of 2's partition holding
    # the whole-mesh oracle's values -- and one holding a wrong value at a shared node is caught
    import numpy as np
    import torch.distributed as dist
    whole_ctx = WholeMeshOracle()
    u1 = bench.seeded_field(box.node_xyz, bench.WORKLOADS["o4s"][0], bench.WORKLOADS["o4s"][1], it)
    whole_ctx.upload(u1, u1 * (1.0 - 1e-3), 0)
    whole_ctx.run(2)
    part, _, _, itp = bench.make_octbox("o4s", 1, 2)
    part.node_ijk, part.start_interfaces = part.node_xyz, itp
    gid = part.gid.astype(np.int64)

    class PartitionContext(WholeMeshOracle):
        wrong_at = -1

        def upload(self, tm1, tm2, step):
            assert np.array_equal(tm1, u1[gid])               # a partition starts from the whole mesh's field

        def run(self, k):
            assert k == 2

        def gather(self, ids):
            a, b = whole_ctx.u1[gid[ids]].copy(), whole_ctx.u2[gid[ids]].copy()
            a[ids == self.wrong_at] += 1e-6
            return a, b

    import os
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        dist.init_process_group("gloo", init_method="file://" + os.path.join(td, "rendezvous"), rank=0, world_size=1)
        try:
            ctx = PartitionContext()
            nwin, nchecked, worst = bench.parity_windows(argparse.Namespace(workload="o4s"), part, ctx, 0, 2)
            assert nwin >= 3 and nchecked > 500 and worst < 1e-12
            sch = part.schedules()
            shared_hanging = np.intersect1d(np.concatenate([m for _, m in sch["dn"]["c"] + sch["dn"]["s"]]), part.dangling[0])
            assert len(shared_hanging) > 0                     # the windows sit where partition and level interfaces meet:
            recs, _ = bench.octree_window_records(argparse.Namespace(workload="o4s"), box, it, 2,
                                                  bench.node_keys(part.node_xyz[shared_hanging]))
            keys = np.concatenate([r[0] for r in recs])
            hit = shared_hanging[np.isin(bench.node_keys(part.node_xyz[shared_hanging]), keys)]
            assert len(hit) > 0                                # ... shared hanging nodes are among the checked ones
            ctx.wrong_at = int(hit[0])
            assert bench.parity_windows(argparse.Namespace(workload="o4s"), part, ctx, 0, 2)[2] > 1e-9
        finally:
            dist.destroy_process_group()
    part.close()

    # --precision f32: a float state stepped by the oracle's float build (compute_adjust on floats) on the rounded n_t rows
    import numpy as np
    nt32 = np.ascontiguousarray(box.ntable, np.float32)

    class FloatMeshOracle(WholeMeshOracle):
        def upload(self, tm1, tm2, step):
            self.u1, self.u2 = np.ascontiguousarray(tm1, np.float32), np.ascontiguousarray(tm2, np.float32)

        def run(self, k):
            o1, o2 = self.u2.copy(), self.u1.copy()
            ho.solver_run(box.lnid, box.etable.copy(), nt32, o1, o2, 0, k, box.dt, dangling=box.dangling)
            self.u1, self.u2 = o2, o1

    nwin, nchecked, worst = bench.parity_windows(argparse.Namespace(workload="o4s", precision="f32"), box, FloatMeshOracle(), 0, 1)
    assert nwin >= 4 and nchecked > 1000 and worst < 1e-6
    box.close()


def test_parity_windows_of_a_single_precision_line(monkeypatch):
    """bench.parity_windows with `--precision f32`: the stand-in keeps a FLOAT state and steps the whole box with the
    oracle's float build on the n_t rows rounded to float (what host._solver_from_desc hands libhq_solver_f32.so); the
    windows must use the same build, tables and start values -- agreement to rounding, tolerance 2e-6 in the line."""
    import argparse
    import numpy as np
    import bench
    from hercules_amd import host
    from oracle import herc_oracle as ho
    monkeypatch.setitem(bench.WORKLOADS, "c1", (128, 32, 32, 62.5, 1e-3, 5.0))
    nx, ny, nz, h, dt, freq = bench.WORKLOADS["c1"]
    box = host.Box(nx, ny, nz, h, dt, freq)
    nt32 = np.ascontiguousarray(box.ntable, np.float32)

    class FloatBoxOracle:
        def set_source(self, ids, F):
            assert len(ids) == 0

        def upload(self, tm1, tm2, step):                       # (capi.Solver.upload casts to the library's hq_real)
            self.u1, self.u2 = np.ascontiguousarray(tm1, np.float32), np.ascontiguousarray(tm2, np.float32)

        def run(self, k):
            o1, o2 = self.u2.copy(), self.u1.copy()
            ho.solver_run(box.lnid, box.etable.copy(), nt32, o1, o2, 0, k, dt)
            self.u1, self.u2 = o2, o1

        def sync(self):
            pass

        def gather(self, ids):
            return self.u1[ids], self.u2[ids]

    args = argparse.Namespace(workload="c1", precision="f32")
    nwin, nchecked, worst = bench.parity_windows(args, box, FloatBoxOracle(), 0, 1)
    assert nwin >= 4 and nchecked > 4 * 11 ** 3 and worst < 1e-6 and bench.parity_tol(args) == 2e-6
    assert bench.parity_tol(argparse.Namespace(workload="c1", precision="f64")) == 1e-9
    # a PARTITION of an octree workload carries no windows in either precision (reported as null)
    box.close()
