"""N > 1 on CPU: world_size-2 (and 3) gloo runs of the partitioned step.  The
product's C host side supplies partitions / ownership / messenger lists, the
oracle does the arithmetic, torch.distributed(gloo) carries the halo records;
the result must equal the oracle's single-rank run."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import herc_oracle as ho

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_step_over_gloo(tmp_path, world):
    nx, ny, nz, nsteps = 16, 8, 8, 15
    h, dt, freq = 20.0, 4e-4, 20.0
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_gloo_worker.py"), str(tmp_path), str(nx), str(ny), str(nz), str(nsteps)]
    out = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         universal_newlines=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:]
    # single-rank oracle
    elem_ijk, lnid, node_ijk = ho.uniform_mesh(nx, ny, nz)
    edata = np.empty((len(lnid), 4), np.float32)
    edata[:] = (h, 6000.0, 3464.0, 2700.0)
    et, nt = ho.solver_init(lnid, edata, ho.face_bits(elem_ijk, nx, ny, nz), len(node_ijk), dt, freq)
    ijk = node_ijk.astype(np.int64)
    gid = (ijk[:, 2] * (ny + 1) + ijk[:, 1]) * (nx + 1) + ijk[:, 0]
    rng = np.random.default_rng(99)
    Ng = (nx + 1) * (ny + 1) * (nz + 1)
    g1 = rng.uniform(-1, 1, (Ng, 3)) * 1e-3
    g2 = g1 + rng.uniform(-1, 1, (Ng, 3)) * 1e-6
    from hercules_amd import host
    b = host.Box(nx, ny, nz, h, dt, freq)
    loaded, pattern = b.point_source(nx * h / 2 + 3.0, ny * h / 2 - 2.0, nz * h / 3, 30.0, 70.0, 10.0)
    rp = b.run_params(loaded=loaded, pattern=pattern, moment=1e13, rise_time=10 * dt)
    F = b.source_table(rp, 0, nsteps)
    o1, o2 = g2[gid].copy(), g1[gid].copy()            # pre-swap arrays
    ho.solver_run(lnid, et, nt, o1, o2, 0, nsteps, dt, loaded_lnid=loaded, forces=F)
    ref1 = np.zeros((Ng, 3)); ref2 = np.zeros((Ng, 3))
    ref1[gid], ref2[gid] = o2, o1                      # post-swap view
    scale = np.abs(ref1).max()
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert np.abs(z["tm1"] - ref1[z["gid"]]).max() <= 1e-12 * scale
        assert np.abs(z["tm2"] - ref2[z["gid"]]).max() <= 1e-12 * scale
    b.close()


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: the parent spawns two fresh rank processes
    (it never re-execs and makes no GPU call), they rendezvous on 127.0.0.1 and rank 0's one JSON
    line says how many ranks it saw.  --dry-launch stops after the rendezvous (no GPU here)."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], cwd=ROOT,
                         env=dict(os.environ, OMP_NUM_THREADS="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         universal_newlines=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d == {"dry_launch": True, "n_gpus": 2, "ranks_seen": 2, "ok": True, "launched_by": "bench.py"}


def test_bench_under_an_external_launcher_uses_its_world_size():
    """The driver's form: torch.distributed.run starts the ranks; bench.py takes RANK / WORLD_SIZE
    from the environment and does not spawn anything itself."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"]
    out = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="1"), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["launched_by"] == "external launcher"
