"""Separate PROCESSES per rank -- what RCCL ranks are -- on one GPU: every rank has its own HIP context, compute and
exchange streams and events, and steps its partition with hq_run (not in lockstep from one thread as hq_group_run
does); the halo records travel through the engine's host-staged transport (hq_comm_init_host: pack on the device,
pinned host buffers, the caller's transport -- here gloo, standing in for the reference's MPI -- and back) or device
to device through the IPC transport (hq_comm_init_ipc: the pack kernel stores every record where the receiving
PROCESS reads it, epoch flags order the two streams), with the
exchange chain on its own stream beside the interior kernels.  Only ncclSend / ncclRecv themselves stay unexercised
on a one-GPU box (RCCL refuses two ranks on one device).  Against the oracle's single-rank run."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import herc_oracle as ho
from tests import helpers as H

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(tmp_path, world, kind, nsteps, env_extra=None, expect_failure=False):
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_hq_rank_worker.py"), str(tmp_path), kind, str(nsteps)]
    out = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         universal_newlines=True, timeout=900)
    if expect_failure:
        assert out.returncode != 0, "the run was expected to fail"
        return out.stdout
    assert out.returncode == 0, out.stdout[-3000:]
    return [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]


@pytest.mark.parametrize("world,overlap,cu_mask,transport",
                         [(2, "1", "0", "host"), (2, "1", "1", "host"),
                          (4, "1", "0", "ipc"), (4, "0", "0", "ipc")])      # (round 6: 4 ranks host-staged and 2 ranks over IPC went -- the suite's time limit)
def test_ranks_in_their_own_processes_on_one_gpu_uniform_box(tmp_path, world, overlap, cu_mask, transport):
    """cu_mask = 1: the compute stream re-created with a CU mask that leaves HQ_RESERVE_CUS CUs to the exchange stream
    (HQ_CU_MASK=1, opt-in: DESIGN.md s6).  transport = ipc: device-to-device between the processes (hq_comm_init_ipc:
    peer stores into IPC-exported receive buffers, epoch flags), no host hop."""
    nx, ny, nz, h, dt, freq, nsteps = 64, 64, 32, 15.0, 3e-4, 30.0, 20
    parts = _launch(tmp_path, world, "box", nsteps, {"HQ_OVERLAP": overlap, "HQ_CU_MASK": cu_mask, "HQ_TEST_TRANSPORT": transport})
    from hercules_amd import host
    b = host.Box(nx, ny, nz, h, dt, freq)
    ijk = b.node_ijk.astype(np.int64)
    gid = (ijk[:, 2] * (ny + 1) + ijk[:, 1]) * (nx + 1) + ijk[:, 0]
    Ng = (nx + 1) * (ny + 1) * (nz + 1)
    rng = np.random.default_rng(4321)
    g1 = rng.uniform(-1, 1, (Ng, 3)) * 1e-3
    g2 = g1 + rng.uniform(-1, 1, (Ng, 3)) * 1e-6
    loaded, pattern = b.point_source(nx * h / 2 + 3.0, ny * h / 2 - 2.0, nz * h / 3, 30.0, 70.0, 10.0)
    rp = b.run_params(loaded=loaded, pattern=pattern, moment=1e13, rise_time=10 * dt)
    F = b.source_table(rp, 0, nsteps)
    o1, o2 = g2[gid].copy(), g1[gid].copy()
    ho.solver_run(b.lnid, b.etable.copy(), b.ntable.copy(), o1, o2, 0, nsteps, dt, loaded_lnid=loaded, forces=F)
    ref1, ref2 = np.zeros((Ng, 3)), np.zeros((Ng, 3))
    ref1[gid], ref2[gid] = o2, o1
    b.close()
    for z in parts:
        assert H.rel_linf(z["tm1"], ref1[z["gid"]]) < 1e-9 and H.rel_linf(z["tm2"], ref2[z["gid"]]) < 1e-9
        assert int(z["brick_nodes"]) > 0 and str(z["kernel"]) == "hq_k_brick"
        assert int(z["transport"]) == (2 if transport == "ipc" else 3)       # hq_info.transport: IPC / host-staged
    if transport == "ipc":
        # first contact of the fine-grained arena: exported here and opened by ANOTHER process (HQ_IPC_COARSE unset)
        assert all(int(z["ipc_arena_coarse"]) == 0 and int(z["ipc_arena_kind"]) == 0 for z in parts)


@pytest.mark.parametrize("transport", ["ipc"])      # (host-staged floats: the fp64 host-staged cases carry the same double records)
def test_float_state_between_processes(tmp_path, transport):
    """libhq_solver_f32.so on ranks in their own processes: the templated forms of the chain's kernels on a float table
    (pack to the peers / to the staging buffers, the interface update with the sharing fused in, the IPC unpack) with the
    check words of hq_options.debug_halo on -- against the oracle's float build on the rounded n_t rows, tolerance as in
    tests/test_gpu_single_precision.py."""
    nx, ny, nz, h, dt, freq, nsteps = 64, 64, 32, 15.0, 3e-4, 30.0, 20
    parts = _launch(tmp_path, 2, "box", nsteps, {"HQ_OVERLAP": "1", "HQ_TEST_TRANSPORT": transport, "HQ_TEST_PRECISION": "f32",
                                                 "HQ_DEBUG_HALO": "1"})
    from hercules_amd import host
    b = host.Box(nx, ny, nz, h, dt, freq)
    ijk = b.node_ijk.astype(np.int64)
    gid = (ijk[:, 2] * (ny + 1) + ijk[:, 1]) * (nx + 1) + ijk[:, 0]
    Ng = (nx + 1) * (ny + 1) * (nz + 1)
    rng = np.random.default_rng(4321)
    g1 = rng.uniform(-1, 1, (Ng, 3)) * 1e-3
    g2 = g1 + rng.uniform(-1, 1, (Ng, 3)) * 1e-6
    loaded, pattern = b.point_source(nx * h / 2 + 3.0, ny * h / 2 - 2.0, nz * h / 3, 30.0, 70.0, 10.0)
    rp = b.run_params(loaded=loaded, pattern=pattern, moment=1e13, rise_time=10 * dt)
    F = b.source_table(rp, 0, nsteps)
    o1, o2 = g2[gid].astype(np.float32), g1[gid].astype(np.float32)
    ho.solver_run(b.lnid, b.etable.copy(), np.ascontiguousarray(b.ntable, np.float32), o1, o2, 0, nsteps, dt, loaded_lnid=loaded, forces=F)
    ref1, ref2 = np.zeros((Ng, 3)), np.zeros((Ng, 3))
    ref1[gid], ref2[gid] = o2, o1
    b.close()
    for z in parts:
        assert z["tm1"].dtype == np.float32 and int(z["debug_halo"]) == 1
        assert H.rel_linf(z["tm1"].astype(np.float64), ref1[z["gid"]]) < 2e-5 and H.rel_linf(z["tm2"].astype(np.float64), ref2[z["gid"]]) < 2e-5
        assert int(z["transport"]) == (2 if transport == "ipc" else 3)


@pytest.mark.parametrize("transport,env", [("ipc", {"HQ_NO_FUSED_SHARE": "1", "HQ_PATCH_MERGE_ROUNDS": "0"}),
                                           ("host", {"HQ_NO_FUSED_SHARE": "1"}),      # (HQ_IPC_COARSE: test_ipc_arena_kinds_between_processes[coarse])
                                           ("ipc", {"HQ_BRICK_BY_COMPONENT": "0", "HQ_BRICK_STREAM": "1"})])
def test_exchange_chain_switches_between_processes(tmp_path, transport, env):
    """The chain's switches that only traces set otherwise: the displacement sharing packed by its own kernel instead of
    by hq_k_interface_update; two patch launches instead of one; a coarse-grained IPC arena (ranks of one device); the
    118-register brick kernel with the bricks on a stream of their own -- two ranks as processes against the oracle."""
    nx, ny, nz, h, dt, freq, nsteps = 64, 64, 32, 15.0, 3e-4, 30.0, 20
    e = {"HQ_OVERLAP": "1", "HQ_CU_MASK": "0", "HQ_TEST_TRANSPORT": transport}
    e.update(env)
    parts = _launch(tmp_path, 2, "box", nsteps, e)
    from hercules_amd import host
    b = host.Box(nx, ny, nz, h, dt, freq)
    ijk = b.node_ijk.astype(np.int64)
    gid = (ijk[:, 2] * (ny + 1) + ijk[:, 1]) * (nx + 1) + ijk[:, 0]
    Ng = (nx + 1) * (ny + 1) * (nz + 1)
    rng = np.random.default_rng(4321)
    g1 = rng.uniform(-1, 1, (Ng, 3)) * 1e-3
    g2 = g1 + rng.uniform(-1, 1, (Ng, 3)) * 1e-6
    loaded, pattern = b.point_source(nx * h / 2 + 3.0, ny * h / 2 - 2.0, nz * h / 3, 30.0, 70.0, 10.0)
    rp = b.run_params(loaded=loaded, pattern=pattern, moment=1e13, rise_time=10 * dt)
    F = b.source_table(rp, 0, nsteps)
    o1, o2 = g2[gid].copy(), g1[gid].copy()
    ho.solver_run(b.lnid, b.etable.copy(), b.ntable.copy(), o1, o2, 0, nsteps, dt, loaded_lnid=loaded, forces=F)
    ref1, ref2 = np.zeros((Ng, 3)), np.zeros((Ng, 3))
    ref1[gid], ref2[gid] = o2, o1
    b.close()
    for z in parts:
        assert H.rel_linf(z["tm1"], ref1[z["gid"]]) < 1e-9 and H.rel_linf(z["tm2"], ref2[z["gid"]]) < 1e-9
        if "HQ_IPC_COARSE" in env:
            assert int(z["ipc_arena_coarse"]) == 1


@pytest.mark.parametrize("transport", ["host", "ipc"])
def test_ranks_in_their_own_processes_on_one_gpu_octree_box(tmp_path, transport):
    """Three ranks of the two-level octree box: hanging nodes shared between ranks, so all four exchanges of a step
    (dangling-node and anchored-node contribution and sharing) go through the host-staged / the IPC transport."""
    nsteps = 12
    parts = _launch(tmp_path, 3, "octree", nsteps, {"HQ_TEST_TRANSPORT": transport})
    ref = H.two_level_mesh(16, 8, 6, 3)
    rng = np.random.default_rng(4321)
    g1 = rng.uniform(-1, 1, (ref["N"], 3)) * 1e-3
    g2 = g1 + rng.uniform(-1, 1, (ref["N"], 3)) * 1e-6
    ho.compute_adjust(g1, 1, ref["dangling"])
    ho.compute_adjust(g2, 1, ref["dangling"])
    o1, o2 = g2.copy(), g1.copy()
    ho.solver_run(ref["lnid"], ref["etable"], ref["ntable"], o1, o2, 0, nsteps, ref["dt"], dangling=ref["dangling"])
    for z in parts:
        assert H.rel_linf(z["tm1"], o2[z["gid"]]) < 1e-9 and H.rel_linf(z["tm2"], o1[z["gid"]]) < 1e-9


@pytest.mark.parametrize("transport", ["ipc", "host"])
def test_the_references_basin_on_five_processes_against_its_own_stripes(tmp_path, transport):
    """BASELINE config 5's mesh class between PROCESSES: the laterally refined basin exactly as the real reference ran
    it on 5 MPI ranks (tests/golden/c5_basin_np5) -- every process builds its partition with the C host from the
    leaves alone, reads the reference's per-rank force file and steps with hq_run; hanging nodes whose anchors live on
    other ranks across x- and y-normal level interfaces, all four exchanges of a step over the IPC / the host-staged
    transport -- against the reference's own per-rank checkpoint stripe of step 100."""
    g = H.load("c5_basin_np5")
    step = int(g["ckpt_steps"][0])
    parts = _launch(tmp_path, 5, "basin", step, {"HQ_TEST_TRANSPORT": transport})
    worst = 0.0
    for r, z in enumerate(parts):
        n = len(z["gid"])
        ref2, ref1 = H.np8_stripe(g, step, r, n)
        scale = max(np.abs(ref1).max(), 1.0)
        worst = max(worst, np.abs(z["tm1"] - ref1).max() / scale, np.abs(z["tm2"] - ref2).max() / scale)
        assert int(z["transport"]) == (2 if transport == "ipc" else 3)
    assert worst < 1e-9
    assert max(np.abs(z["tm1"]).max() for z in parts) > 100.0


@pytest.mark.parametrize("transport", ["ipc", "host"])
def test_halo_debug_mode_between_processes(tmp_path, transport):
    """HQ_DEBUG_HALO=1 (the reference's -DDEBUG exchange, psolve.c:5002-5007, 5058-5069) on the transports that run
    between processes: every record of all four exchanges travels with a check word -- its node's global identity mixed
    with the number of the exchange and the record's own values (an id arena beside the IPC record arena; a second
    message per neighbour on the host-staged transport).  Three ranks of the two-level octree box: a correct run passes
    and equals the oracle; with two records of one rank's c-list swapped the owner's check fails and hq_sync says so."""
    nsteps = 8
    env = {"HQ_TEST_TRANSPORT": transport, "HQ_DEBUG_HALO": "1"}
    parts = _launch(tmp_path, 3, "octree", nsteps, env)
    ref = H.two_level_mesh(16, 8, 6, 3)
    rng = np.random.default_rng(4321)
    g1 = rng.uniform(-1, 1, (ref["N"], 3)) * 1e-3
    g2 = g1 + rng.uniform(-1, 1, (ref["N"], 3)) * 1e-6
    ho.compute_adjust(g1, 1, ref["dangling"])
    ho.compute_adjust(g2, 1, ref["dangling"])
    o1, o2 = g2.copy(), g1.copy()
    ho.solver_run(ref["lnid"], ref["etable"], ref["ntable"], o1, o2, 0, nsteps, ref["dt"], dangling=ref["dangling"])
    for z in parts:
        assert int(z["debug_halo"]) == 1
        assert H.rel_linf(z["tm1"], o2[z["gid"]]) < 1e-9 and H.rel_linf(z["tm2"], o1[z["gid"]]) < 1e-9
    bad = tmp_path / "bad"
    bad.mkdir()
    log = _launch(bad, 3, "octree", 2, dict(env, HQ_TEST_SWAP_RANK="1"), expect_failure=True)
    errs = [f.read_text() for f in bad.glob("rank*.err")]
    assert errs and all("HQ_DEBUG_HALO" in e for e in errs), log[-2000:]


@pytest.mark.parametrize("kind", ["uncached", "coarse"])
def test_ipc_arena_kinds_between_processes(tmp_path, kind):
    """The IPC receive arena as uncached device memory (the fallback where a runtime will not export a fine-grained
    allocation) and as coarse-grained memory (ranks of one device only): two ranks as processes against the oracle."""
    nx, ny, nz, h, dt, freq, nsteps = 64, 64, 32, 15.0, 3e-4, 30.0, 12
    parts = _launch(tmp_path, 2, "box", nsteps, {"HQ_OVERLAP": "1", "HQ_CU_MASK": "0", "HQ_TEST_TRANSPORT": "ipc", "HQ_IPC_ARENA": kind})
    from hercules_amd import host
    b = host.Box(nx, ny, nz, h, dt, freq)
    ijk = b.node_ijk.astype(np.int64)
    gid = (ijk[:, 2] * (ny + 1) + ijk[:, 1]) * (nx + 1) + ijk[:, 0]
    Ng = (nx + 1) * (ny + 1) * (nz + 1)
    rng = np.random.default_rng(4321)
    g1 = rng.uniform(-1, 1, (Ng, 3)) * 1e-3
    g2 = g1 + rng.uniform(-1, 1, (Ng, 3)) * 1e-6
    loaded, pattern = b.point_source(nx * h / 2 + 3.0, ny * h / 2 - 2.0, nz * h / 3, 30.0, 70.0, 10.0)
    rp = b.run_params(loaded=loaded, pattern=pattern, moment=1e13, rise_time=10 * dt)
    F = b.source_table(rp, 0, nsteps)
    o1, o2 = g2[gid].copy(), g1[gid].copy()
    ho.solver_run(b.lnid, b.etable.copy(), b.ntable.copy(), o1, o2, 0, nsteps, dt, loaded_lnid=loaded, forces=F)
    ref1, ref2 = np.zeros((Ng, 3)), np.zeros((Ng, 3))
    ref1[gid], ref2[gid] = o2, o1
    b.close()
    for z in parts:
        assert H.rel_linf(z["tm1"], ref1[z["gid"]]) < 1e-9 and H.rel_linf(z["tm2"], ref2[z["gid"]]) < 1e-9
        assert int(z["ipc_arena_kind"]) == {"uncached": 1, "coarse": 2}[kind]
