"""round 6: eight partitions built by the C host side (messenger lists in the reference's order) on one GPU against the stripes
the reference's 8 MPI ranks wrote -- double and float."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import helpers as H
from hercules_amd import host, capi

for name, sf, prec in (("c5_two_level_np8", 8, "f64"), ("c5_two_level_np8_f32", 4, "f32"), ("c5_basin_np8", 8, "f64"), ("c5_gradient_np8", 8, "f64")):
    g = H.load(name); base = H.load(str(g["base"])); P = int(g["nranks"])
    et = base["elem_ticks"]; edge = et[:, 7, 0] - et[:, 0, 0]; mat = base["mat_vs_vp_rho"]
    edata = np.empty((len(et), 4), np.float32)
    edata[:, 0] = (edge * (1000.0 / 2 ** 30)).astype(np.float32)
    edata[:, 1], edata[:, 2], edata[:, 3] = mat[:, 1], mat[:, 0], mat[:, 2]
    boxes = [host.OctBox.from_leaves(et[:, 0, :], edge, edata, H.C1_FAR_TICKS, 1e-3, float(base["freq"]), rank=r, nranks=P, solver_float=sf) for r in range(P)]
    solvers = [b.create_solver(precision=prec) for b in boxes]
    for r, s in enumerate(solvers):
        if len(g["loaded_lnid_%d" % r]):
            s.set_source(g["loaded_lnid_%d" % r], g["forces_%d" % r])
    capi.group_link(solvers)
    done, out = 0, []
    for step in g["ckpt_steps"]:
        capi.group_run(solvers, int(step) - done); done = int(step)
        stripes = [H.np8_stripe(g, step, r, boxes[r].N) for r in range(P)]
        scale = max(float(np.abs(ref1).max()) for _, ref1 in stripes)
        e = 0.0
        for (ref2, ref1), s in zip(stripes, solvers):
            tm1, tm2 = s.download()
            e = max(e, np.abs(tm1.astype(np.float64) - ref1).max(), np.abs(tm2.astype(np.float64) - ref2).max())
        out.append("%d: %.2e" % (done, e / scale))
    print("%-22s %s state, %d partitions: %s" % (name, prec, P, "  ".join(out)))
    for s in solvers: s.close()
    for b in boxes: b.close()
