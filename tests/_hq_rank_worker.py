"""Worker of tests/test_gpu_multiprocess.py: ONE RANK of a partitioned run in its own process -- its own HIP
context, streams and events -- on the GPU all ranks of the test share.  The halo records travel through the
engine's host-staged transport (hq_comm_init_host) and torch.distributed's gloo, or (HQ_TEST_TRANSPORT=ipc) device to
device through the IPC transport (hq_comm_ipc_export / hq_comm_init_ipc) with gloo only for the set-up all-gather;
gloo stands in for the MPI of the reference's world (psolve.c:7344-7389 `mpiexec -np N`; schedule_senddata psolve.c:4945-5079).  Everything else is
the product path: C host side for the partition, hq_create / hq_set_source / hq_run on the device.
Writes this rank's final fields to <outdir>/rank<r>.npz."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import hercules_amd as ha               # noqa: E402
from hercules_amd import host           # noqa: E402


def gloo_exchange(recvs, sends, tag):
    """schedule_senddata's MPI_Irecv / MPI_Isend / MPI_Waitall (psolve.c:5013-5033) over gloo."""
    reqs = [dist.irecv(torch.from_numpy(buf), src=int(peer), tag=int(tag)) for peer, buf in recvs]
    reqs += [dist.isend(torch.from_numpy(buf), dst=int(peer), tag=int(tag)) for peer, buf in sends]
    for r in reqs:
        r.wait()


def main():
    outdir, kind, nsteps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(4321)
    if kind == "box":
        nx, ny, nz, h, dt, freq = 64, 64, 32, 15.0, 3e-4, 30.0
        b = host.Box(nx, ny, nz, h, dt, freq, rank=rank, nranks=world)
        ijk = b.node_ijk.astype(np.int64)
        gid = (ijk[:, 2] * (ny + 1) + ijk[:, 1]) * (nx + 1) + ijk[:, 0]
        Ng = (nx + 1) * (ny + 1) * (nz + 1)
        loaded, pattern = b.point_source(nx * h / 2 + 3.0, ny * h / 2 - 2.0, nz * h / 3, 30.0, 70.0, 10.0)
        rp = b.run_params(loaded=loaded, pattern=pattern, moment=1e13, rise_time=10 * dt)
        F = b.source_table(rp, 0, nsteps) if len(loaded) else None
    elif kind == "basin":
        # the reference's laterally refined basin as it ran it on `world` MPI ranks (tests/golden/c5_basin_np<world>):
        # this rank's partition built by the C host from the leaves alone, the reference's own per-rank force file,
        # zero start -- the parent compares with the reference's per-rank checkpoint stripes
        from tests import helpers as H
        g, base = H.load("c5_basin_np%d" % world), H.load("c5_basin")
        et = base["elem_ticks"]
        edge = et[:, 7, 0] - et[:, 0, 0]
        mat = base["mat_vs_vp_rho"]
        edata = np.empty((len(et), 4), np.float32)
        edata[:, 0] = (edge * (1000.0 / 2 ** 30)).astype(np.float32)
        edata[:, 1], edata[:, 2], edata[:, 3] = mat[:, 1], mat[:, 0], mat[:, 2]
        b = host.OctBox.from_leaves(et[:, 0, :], edge, edata, H.C1_FAR_TICKS, 1e-3, float(base["freq"]), rank=rank, nranks=world)
        gid, Ng = b.gid.astype(np.int64), int(base["total_nodes"])
        loaded, F = g["loaded_lnid_%d" % rank], g["forces_%d" % rank]
        if len(loaded) == 0:
            F = None
    else:                                                   # two-level octree box: all four exchanges of a step
        from tests import helpers as H
        ref = H.two_level_mesh(16, 8, 6, 3)
        b = host.OctBox(16, 8, 6, 3, 31.25, ref["dt"], 5.0, rank=rank, nranks=world)
        gid, Ng, loaded, F = b.gid.astype(np.int64), ref["N"], np.zeros(0, np.int32), None
    g1 = rng.uniform(-1, 1, (Ng, 3)) * 1e-3
    g2 = g1 + rng.uniform(-1, 1, (Ng, 3)) * 1e-6
    if kind == "basin":
        g1[:], g2[:] = 0.0, 0.0
    elif kind != "box":
        from oracle import herc_oracle as ho                # (checker-side helper: hanging rows = means of their anchors)
        ho.compute_adjust(g1, 1, ref["dangling"])
        ho.compute_adjust(g2, 1, ref["dangling"])
    if os.environ.get("HQ_TEST_SWAP_RANK") == str(rank):
        # a deliberately WRONG schedule on this rank: two records of one c-list messenger swapped, so that the owner adds
        # them to the wrong nodes -- what HQ_DEBUG_HALO exists to catch (the parent expects the run to fail)
        sch = b.schedules()
        k = next(i for i, (_, m) in enumerate(sch["an"]["c"]) if len(m) > 1)
        sch["an"]["c"][k][1][[0, 1]] = sch["an"]["c"][k][1][[1, 0]]
        s = ha.Solver(b.lnid, b.etable, b.ntable, b.dt, tm1=g1[gid], tm2=g2[gid], node_xyz=b.node_xyz, dangling=b.dangling,
                      an_sched=sch["an"], dn_sched=sch["dn"], rank=rank, nranks=world, variant=ha.HQ_VARIANT_PATCH)
    else:
        # HQ_TEST_PRECISION=f32: libhq_solver_f32.so -- a float state on every rank, the records still travel as doubles
        s = b.create_solver(variant=ha.HQ_VARIANT_PATCH, tm1=g1[gid], tm2=g2[gid], precision=os.environ.get("HQ_TEST_PRECISION", "f64"))
    if os.environ.get("HQ_TEST_TRANSPORT", "host") == "ipc":
        # device-to-device between the processes: blobs all-gathered over gloo (MPI_Allgather in the reference's world)
        mine = torch.frombuffer(bytearray(s.comm_ipc_export()), dtype=torch.uint8)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        s.comm_init_ipc([bytes(t.numpy().tobytes()) for t in every])
    else:
        s.comm_init_host(gloo_exchange)
    if F is not None:
        s.set_source(loaded, F)
    dist.barrier()
    failed = None
    try:
        s.run(nsteps)
        s.sync()
    except ha.HqError as e:                                 # e.g. HQ_DEBUG_HALO's report: every rank still leaves in step
        failed = str(e)
        open(os.path.join(outdir, "rank%d.err" % rank), "w").write(failed)
    tm1, tm2 = s.download()
    info = s.info()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), gid=gid, tm1=tm1, tm2=tm2, brick_nodes=info["brick_nodes"],
             kernel=s.dominant_kernel(), transport=info["transport"], ipc_arena_coarse=info["ipc_arena_coarse"],
             ipc_arena_kind=info["ipc_arena_kind"], debug_halo=info["debug_halo"])
    s.close()
    b.close()
    dist.barrier()
    dist.destroy_process_group()
    if failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
