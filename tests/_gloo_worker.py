"""Worker of tests/test_distributed_gloo.py (one process per partition, gloo).

The partition, ownership and messenger lists come from the PRODUCT's C host
side (hercules_amd/csrc/hq_host.c); the element/node arithmetic is the oracle's
(CPU); the halo exchange follows schedule_senddata (psolve.c:4945-5079) over
torch.distributed.  Writes this rank's final fields to <outdir>/rank<r>.npz."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hercules_amd import host            # noqa: E402
from oracle import herc_oracle as ho     # noqa: E402


def exchange(sched, table, contribution):
    snd = sched["c"] if contribution else sched["s"]
    rcv = sched["s"] if contribution else sched["c"]
    reqs, bufs = [], []
    for proc, mapping in rcv:
        t = torch.empty((len(mapping), 3), dtype=torch.float64)
        bufs.append((mapping, t))
        reqs.append(dist.irecv(t, src=proc))
    for proc, mapping in snd:
        reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(table[mapping])), dst=proc))
    for r in reqs:
        r.wait()
    for mapping, t in bufs:
        if contribution:
            np.add.at(table, mapping, t.numpy())
        else:
            table[mapping] = t.numpy()


def main():
    outdir, nx, ny, nz, nsteps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    h, dt, freq = 20.0, 4e-4, 20.0
    b = host.Box(nx, ny, nz, h, dt, freq, rank=rank, nranks=world)
    sched = b.schedule()
    lnid, et, nt = b.lnid.copy(), b.etable.copy(), b.ntable.copy()
    N = len(nt)
    ijk = b.node_ijk.astype(np.int64)
    gid = (ijk[:, 2] * (ny + 1) + ijk[:, 1]) * (nx + 1) + ijk[:, 0]
    rng = np.random.default_rng(99)
    Ng = (nx + 1) * (ny + 1) * (nz + 1)
    g1 = rng.uniform(-1, 1, (Ng, 3)) * 1e-3
    g2 = g1 + rng.uniform(-1, 1, (Ng, 3)) * 1e-6
    tm1, tm2 = g1[gid].copy(), g2[gid].copy()          # post-swap view: tm1 = u(t)
    loaded, pattern = b.point_source(nx * h / 2 + 3.0, ny * h / 2 - 2.0, nz * h / 3, 30.0, 70.0, 10.0)
    rp = b.run_params(loaded=loaded, pattern=pattern, moment=1e13, rise_time=10 * dt)
    F = b.source_table(rp, 0, nsteps) if len(loaded) else None
    K1, K2 = ho.compute_K()
    L = ho.lib()
    E = len(lnid)
    force = np.zeros((N, 3))
    c64 = ho.ctypes.c_int64
    for step in range(nsteps):
        if F is not None:
            L.ho_addforce_source(len(loaded), ho._p(loaded), ho._p(np.ascontiguousarray(F[step])),
                                 ho.ctypes.c_double(dt * dt), ho._p(force))
        L.ho_addforce_effective(c64(E), ho._p(lnid), ho._p(et), ho._p(tm1), ho._p(force), 1)
        L.ho_damping_addforce(c64(E), ho._p(lnid), ho._p(et), ho._p(tm1), ho._p(tm2), ho._p(K1), ho._p(K2),
                              ho._p(force), 1)
        exchange(sched, force, True)                                   # psolve.c:4301
        L.ho_compute_displacement(c64(N), ho._p(nt), ho._p(tm1), ho._p(tm2), ho._p(force), None)
        exchange(sched, tm2, False)                                    # psolve.c:4312
        tm1, tm2 = tm2, tm1
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), gid=gid, tm1=tm1, tm2=tm2, owner=b.owner)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
