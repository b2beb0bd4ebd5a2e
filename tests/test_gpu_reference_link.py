"""The drop-in boundary exercised by the reference itself: oracle/_ref/psolve_hq is the REAL CMU-Quake/hercules
psolve -- its main(), mesher (octor on the example's material database), solver_init, source and station code --
with INTEGRATION.md's stub applied to a scratch copy of psolve.c at build time (oracle/build_ref_hq.sh: hq_attach
behind solver_init, hq_steps in place of solver_run()'s physics + communication block psolve.c:4286-4316,
hq_refresh_host where the outputs read tm1 / tm2) and linked against libhq_solver.so.

Run on examples/simple (16 x 16 x 8 elements, 1 rank; mpiexec starts the program before anything touches the GPU):
* its checkpoints equal the ones the unmodified reference wrote for the golden fixture (steps 400 / 800 of the
  1000-step run) to the parity bar 1e-9;
* its station files equal those of the unmodified psolve run beside it on the host cores, line for line to the
  printed precision;
* and (20 000 steps) the station traces the reference SHIPS in examples/simple/expected-out.
Needs the two binaries (built in the build container, they travel with the repo snapshot) and the image's MPICH."""
import os
import re
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import ref_baseline as rb
from tests import helpers as H

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PSOLVE_HQ = os.path.join(ROOT, "oracle", "_ref", "psolve_hq")
MPI = os.environ.get("HERC_MPI_DIR", "/opt/conda")


def _setkey(t, key, val):
    assert re.search(r"(?m)^%s\s*=.*$" % re.escape(key), t), key
    return re.sub(r"(?m)^%s\s*=.*$" % re.escape(key), "%s = %s" % (key, val), t)


def _params(end_time, ckpt_rate, station_rate):
    t = rb.PARAMS.format(freq=5.0, dt=0.001, end_time=repr(end_time))
    t = _setkey(t, "checkpointing_rate", ckpt_rate)
    t = _setkey(t, "number_output_stations", len(H.C1_STATIONS))
    t = _setkey(t, "output_stations_print_rate", station_rate)
    t = t.replace("output_stations =\n500.0 500.0 100.0\n",
                  "output_stations =\n" + "".join("%.1f %.1f %.5f\n" % s for s in H.C1_STATIONS))
    return t


def _run(binary, params, timeout=900, nranks=1, env_extra=None):
    run = tempfile.mkdtemp(prefix="herc_hq_", dir="/tmp")
    shutil.copy(os.path.join(rb.INPUTS, "simple_case.e"), run)
    shutil.copytree(os.path.join(rb.INPUTS, "sourcefiles"), os.path.join(run, "sourcefiles"))
    for d in ("checkpoints", "planes", "srctmp", "stations"):
        os.makedirs(os.path.join(run, "out", d))
    open(os.path.join(run, "parameters.in"), "w").write(params)
    # the system's libstdc++ (libhq_solver.so needs it) ahead of the older one beside the image's MPICH
    syslib = "/usr/lib/x86_64-linux-gnu"
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0",
               LD_LIBRARY_PATH=":".join([syslib, os.path.join(MPI, "lib"), os.environ.get("LD_LIBRARY_PATH", "")]))
    env.update(env_extra or {})
    out = subprocess.run([os.path.join(MPI, "bin", "mpiexec"), "-np", str(nranks), binary, "parameters.in"], cwd=run, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=timeout)
    assert out.returncode == 0, out.stdout[-2000:]
    return run, out.stdout


def _stations(run, n=5, ncol=4):
    def one(i):
        rows = [l.split() for l in open(os.path.join(run, "out", "stations", "station.%d" % i)).read().splitlines()
                if l.strip() and not l.lstrip().startswith("#")]
        return np.array([[float(v) for v in r[:ncol]] for r in rows])
    return np.stack([one(i) for i in range(n)])


def _checkpoints(run):
    out = {}
    for k in (0, 1):
        b = open(os.path.join(run, "out", "checkpoints", "checkpoint.out%d" % k), "rb").read()
        groupsize, step, nmax = [int(v) for v in np.frombuffer(b[:12], "<i4")]
        assert groupsize == 1
        tm2 = np.frombuffer(b[12:12 + nmax * 24], "<f8").reshape(nmax, 3)
        tm1 = np.frombuffer(b[12 + nmax * 24:12 + 2 * nmax * 24], "<f8").reshape(nmax, 3)
        out[step] = (tm2, tm1)
    return out


def _rank_stripes(run):
    """{step: [(tm2, tm1) of every rank]} of a multi-rank run's checkpoint files (io_checkpoint.c:76-112)."""
    out = {}
    for k in (0, 1):
        b = open(os.path.join(run, "out", "checkpoints", "checkpoint.out%d" % k), "rb").read()
        groupsize, step, nmax = [int(v) for v in np.frombuffer(b[:12], "<i4")]
        ranks = []
        for r in range(groupsize):
            off = 12 + 2 * r * nmax * 24
            raw = np.frombuffer(b[off:off + 2 * nmax * 24].ljust(2 * nmax * 24, b"\0"), "<f8")
            ranks.append((raw[:3 * nmax].reshape(nmax, 3), raw[3 * nmax:].reshape(nmax, 3)))
        out[step] = ranks
    return out


needs_binaries = pytest.mark.skipif(
    not (os.path.exists(PSOLVE_HQ) and rb.available()),
    reason="oracle/_ref/psolve_hq, oracle/_ref/psolve (oracle/build_ref*.sh, build container) or mpiexec missing")


@needs_binaries
def test_the_reference_program_on_the_library_reproduces_its_own_checkpoints_and_stations():
    g = H.load("c1_short")
    params = _params(float(g["end_time"]), 400, 1)
    run_hq, log = _run(PSOLVE_HQ, params)
    run_ref, _ = _run(rb.PSOLVE, params)
    try:
        assert "Total elements:" in log and int(re.search(r"Total elements:\s+(\d+)", log).group(1)) == 2048
        ck, ck_ref = _checkpoints(run_hq), _checkpoints(run_ref)
        assert sorted(ck) == sorted(ck_ref) == [int(s) for s in g["ckpt_steps"]]
        for k, step in enumerate(g["ckpt_steps"]):
            tm2, tm1 = ck[int(step)]
            # the golden fixture (written by the unmodified reference in the build container) ...
            assert H.rel_linf(tm1, g["ckpt_tm1"][k]) < 1e-9 and H.rel_linf(tm2, g["ckpt_tm2"][k]) < 1e-9
            # ... and the unmodified reference run just now on this box's host cores
            assert H.rel_linf(tm1, ck_ref[int(step)][1]) < 1e-9 and H.rel_linf(tm2, ck_ref[int(step)][0]) < 1e-9
            assert np.abs(tm1).max() > 0
        st, st_ref = _stations(run_hq), _stations(run_ref)
        assert st.shape == st_ref.shape == (5, 1000, 4)
        scale = np.abs(st_ref[:, :, 1:]).max()
        assert scale > 0 and np.abs(st - st_ref).max() <= 2e-6 * scale           # %e text: 6 digits behind the point
        assert np.abs(st - g["stations"]).max() <= 2e-6 * scale
    finally:
        shutil.rmtree(run_hq, ignore_errors=True)
        shutil.rmtree(run_ref, ignore_errors=True)


@needs_binaries
def test_the_reference_program_on_the_library_matches_the_shipped_station_traces():
    """The whole examples/simple run (20 000 steps) through psolve_hq against the station traces the reference ships
    (examples/simple/expected-out/stations; every 20th line and the first 400 are in the fixture)."""
    g = H.load("c1_full")
    run_hq, _ = _run(PSOLVE_HQ, _params(float(g["end_time"]), 100000000, 1), timeout=1500)
    try:
        st = _stations(run_hq)
        assert st.shape == (5, 20000, 4)
        exp = g["expected_every20"]
        mine = st[:, ::20, :][:, :exp.shape[1], :]
        scale = np.abs(exp[:, :, 1:]).max()
        assert scale > 1000.0 and np.abs(mine[:, :, 1:] - exp[:, :, 1:]).max() <= 2e-6 * scale
        head = g["expected_head"]
        assert np.abs(st[:, :head.shape[1], 1:] - head[:, :, 1:]).max() <= 2e-6 * np.abs(head[:, :, 1:]).max()
    finally:
        shutil.rmtree(run_hq, ignore_errors=True)


@needs_binaries
@pytest.mark.parametrize("transport", ["mpi", "ipc"])
def test_the_reference_program_on_eight_ranks_shares_one_gpu(transport):
    """mpiexec -np 8 psolve_hq on examples/simple, all eight ranks on this box's one GPU: the reference's own octor
    partition, schedule_build messengers (psolve.c:4704-4863) and mpiexec world (psolve.c:7344-7389) feed hq_desc on
    every rank; the halo records travel on the reference's own MPI_Irecv / MPI_Isend / MPI_Waitall (HQ_TRANSPORT=mpi:
    hq_comm_init_host with the stub's callback, psolve.c:5013-5033) or device to device between the eight processes
    (HQ_TRANSPORT=ipc).  Against the per-rank checkpoint stripes and the station traces of the unmodified reference's
    8-rank run (tests/golden/c1_np8.npz)."""
    g = H.load("c1_np8")
    # (stations every 10 steps: between outputs the eight processes enqueue whole batches instead of synchronising on
    #  the one GPU they time-slice at every step -- 27 s -> a few; the every-step cadence is the one-rank test's)
    run_hq, log = _run(PSOLVE_HQ, _params(float(g["end_time"]), 400, 10), nranks=8, env_extra={"HQ_TRANSPORT": transport})
    try:
        ck = _rank_stripes(run_hq)
        assert sorted(ck) == [int(s) for s in g["ckpt_steps"]]
        for step in g["ckpt_steps"]:
            assert len(ck[int(step)]) == 8
            for r in range(8):
                ref2, ref1 = g["ckpt%d_tm2_%d" % (int(step), r)], g["ckpt%d_tm1_%d" % (int(step), r)]
                tm2, tm1 = ck[int(step)][r]
                n = len(ref1)
                scale = max(np.abs(ref1).max(), 1e-300)
                assert scale > 0 and np.abs(tm1[:n] - ref1).max() <= 1e-9 * scale and np.abs(tm2[:n] - ref2).max() <= 1e-9 * scale
        st = _stations(run_hq)
        assert g["stations"].shape == (5, 1000, 4) and st.shape == (5, 100, 4)
        scale = np.abs(g["stations"][:, :, 1:]).max()
        assert scale > 0 and np.abs(st - g["stations"][:, ::10, :]).max() <= 2e-6 * scale
        # the library's share of print_timing_stat (hq_print_timing in the stub): the device-side split of a step
        assert "Device timers (libhq_solver" in log and "exchange chain" in log
    finally:
        shutil.rmtree(run_hq, ignore_errors=True)


@needs_binaries
def test_the_reference_program_reads_the_device_at_its_output_cadence_only():
    """Stations every 10 steps, a checkpoint every 400: between outputs psolve_hq enqueues whole batches (hq_steps with
    n = steps to the next due output) and refreshes only what the due output reads (the stations' 8 nodes each through
    hq_gather; the whole field at checkpoints) -- the station lines and checkpoints must equal the every-step run's."""
    g = H.load("c1_short")
    run_hq, _ = _run(PSOLVE_HQ, _params(float(g["end_time"]), 400, 10))
    try:
        ck = _checkpoints(run_hq)
        for k, step in enumerate(g["ckpt_steps"]):
            tm2, tm1 = ck[int(step)]
            assert H.rel_linf(tm1, g["ckpt_tm1"][k]) < 1e-9 and H.rel_linf(tm2, g["ckpt_tm2"][k]) < 1e-9
        st = _stations(run_hq)
        assert st.shape == (5, 100, 4)
        ref = g["stations"][:, ::10, :]
        scale = np.abs(g["stations"][:, :, 1:]).max()
        assert np.abs(st - ref).max() <= 2e-6 * scale
    finally:
        shutil.rmtree(run_hq, ignore_errors=True)


@needs_binaries
def test_station_accelerations_at_steps_that_also_write_a_checkpoint():
    """print_station_accelerations = yes with stations every 10 steps and a checkpoint every 100: at steps 100, 200, ...
    the whole field comes back for the checkpoint (hq_download: tm1 / tm2) AND the stations' tm3 rows must still come
    from the device (hq_gather3) -- interpolate_station_displacements reads them (psolve.c:6775-6777).  All ten columns
    against the unmodified reference's run with the same switches (tests/golden/c1_stations_va.npz)."""
    g = H.load("c1_stations_va")
    t = _params(float(g["end_time"]), 100, 10)
    t = _setkey(t, "print_station_velocities", "yes")
    t = _setkey(t, "print_station_accelerations", "yes")
    run_hq, _ = _run(PSOLVE_HQ, t)
    try:
        st = _stations(run_hq, ncol=10)
        ref = g["stations"][:, ::10, :]
        assert st.shape == ref.shape == (5, 60, 10)
        for q in range(3):                                  # displacement, velocity, acceleration: each in its own scale
            cols = slice(1 + 3 * q, 4 + 3 * q)
            scale = np.abs(g["stations"][:, :, cols]).max()
            assert scale > 0 and np.abs(st[:, :, cols] - ref[:, :, cols]).max() <= 2e-6 * scale
        assert np.abs(ref[:, 10::10, 7:]).max() > 1.0        # the lines written at checkpoint steps carry real accelerations
    finally:
        shutil.rmtree(run_hq, ignore_errors=True)


PSOLVE_HQ_F32 = os.path.join(ROOT, "oracle", "_ref", "psolve_hq_f32")


@pytest.mark.skipif(not (os.path.exists(PSOLVE_HQ_F32) and rb.available()),
                    reason="oracle/_ref/psolve_hq_f32 (oracle/build_ref_hq.sh, build container) or mpiexec missing")
def test_the_single_precision_reference_program_on_the_float_library():
    """The reference built with -DSINGLE_PRECISION_SOLVER (psolve.h:60-64) with the same stub, compiled with
    -DHQ_SINGLE_PRECISION_SOLVER and linked against libhq_solver_f32.so: its float arrays go through the C-ABI as they are
    (hq_real = solver_float; hq_attach checks the pairing).  Checkpoints (rows of three floats, io_checkpoint.c:98-112) and
    station files against what the unmodified float reference wrote for tests/golden/c1_f32.npz: within the tolerance the
    f32 dtype states (tests/test_gpu_single_precision.py: 2e-5 over this 800-step run)."""
    g = H.load("c1_f32")
    run_hq, log = _run(PSOLVE_HQ_F32, _params(float(g["end_time"]), 400, 1))
    try:
        assert int(re.search(r"Total elements:\s+(\d+)", log).group(1)) == 2048
        seen = []
        for k in (0, 1):
            b = open(os.path.join(run_hq, "out", "checkpoints", "checkpoint.out%d" % k), "rb").read()
            groupsize, step, nmax = [int(v) for v in np.frombuffer(b[:12], "<i4")]
            assert groupsize == 1 and len(b) == 12 + 2 * nmax * 12              # float rows
            tm2 = np.frombuffer(b[12:12 + nmax * 12], "<f4").reshape(nmax, 3).astype(np.float64)
            tm1 = np.frombuffer(b[12 + nmax * 12:], "<f4").reshape(nmax, 3).astype(np.float64)
            i = list(g["ckpt_steps"]).index(step)
            assert H.rel_linf(tm1, g["ckpt_tm1"][i].astype(np.float64)) < 2e-5
            assert H.rel_linf(tm2, g["ckpt_tm2"][i].astype(np.float64)) < 2e-5
            seen.append(step)
        assert sorted(seen) == [int(s) for s in g["ckpt_steps"]]
        st = _stations(run_hq)
        scale = np.abs(g["stations"][:, :, 1:]).max()
        assert st.shape == g["stations"].shape and np.abs(st - g["stations"]).max() <= 2e-5 * scale
    finally:
        shutil.rmtree(run_hq, ignore_errors=True)
