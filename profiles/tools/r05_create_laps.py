import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ["HQ_PATCH_VERBOSE"] = "2"
import numpy as np
import bench
from hercules_amd import host
nx, ny, nz, h, dt, freq = bench.WORKLOADS["c3"]
for r in (0, 5):
    t = time.time(); b = host.Box(nx, ny, nz, h, dt, freq, rank=r, nranks=8); print("box rank", r, "%.1f s" % (time.time() - t), file=sys.stderr)
    u = np.zeros((len(b.node_xyz) if hasattr(b, "node_xyz") else len(b.node_ijk), 3))
    t = time.time(); s = b.create_solver(tm1=u, tm2=u); print("hq_create rank", r, "%.1f s" % (time.time() - t), s.info()["npatches"], file=sys.stderr)
    s.close(); b.close()
t = time.time(); b = bench.make_octbox("o3", 3, 8)[0]; print("o3 rank 3 mesh %.1f s" % (time.time() - t), file=sys.stderr)
u = np.zeros((len(b.node_xyz) if hasattr(b, "node_xyz") else len(b.node_ijk), 3))
t = time.time(); s = b.create_solver(tm1=u, tm2=u); print("o3 hq_create rank 3 %.1f s" % (time.time() - t), s.info()["npatches"], file=sys.stderr)
