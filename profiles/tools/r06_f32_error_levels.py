"""round 6: how far is the float library from the float reference's own checkpoints when it steps the float build's own rows?
(the figures behind the tolerances in tests/test_gpu_single_precision.py)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hercules_amd as ha
from tests import helpers as H

def ticks(node_ijk, edge=1 << 26):
    return (np.asarray(node_ijk, np.int64) * edge).astype(np.int32)

g = H.load("c1_f32")
for real, label in ((np.float32, "the float build's rows"), (np.float64, "the double build's rows, rounded")):
    p = H.c1_problem("rayleigh", real=real)
    nt = p["ntable"] if real == np.float32 else np.ascontiguousarray(p["ntable"], np.float32)
    for variant, vn in ((ha.HQ_VARIANT_PATCH, "patch"), (ha.HQ_VARIANT_SCATTER, "scatter")):
        s = ha.Solver(p["lnid"], p["etable"], nt, p["dt"], node_xyz=ticks(p["node_ijk"]), variant=variant, precision="f32")
        s.set_source(g["loaded_lnid"], g["forces"])
        done, out = 0, []
        for k, step in enumerate(g["ckpt_steps"]):
            s.run(int(step) - done); done = int(step)
            tm1, tm2 = s.download()
            out.append("%d: %.2e" % (int(step), max(H.rel_linf(tm1.astype(np.float64), g["ckpt_tm1"][k].astype(np.float64)),
                                                      H.rel_linf(tm2.astype(np.float64), g["ckpt_tm2"][k].astype(np.float64)))))
        s.close()
        print("c1 (examples/simple, one rank), %s, %s: %s" % (label, vn, "  ".join(out)))
p32 = H.c5_problem("c5_two_level_f32", real=np.float32)
p64 = H.c5_problem("c5_two_level_f32", real=np.float64)
g = p32["golden"]
for nt, label in ((p32["ntable"], "the float build's rows"), (np.ascontiguousarray(p64["ntable"], np.float32), "the double build's rows, rounded")):
    s = ha.Solver(p32["lnid"], p32["etable"], nt, p32["dt"], dangling=p32["dangling"],
                  node_xyz=(p32["node_q"].astype(np.int64) * p32["emin"]).astype(np.int32), variant=ha.HQ_VARIANT_PATCH, precision="f32")
    s.set_source(g["loaded_lnid"], g["forces"])
    done, out = 0, []
    for k, step in enumerate(g["ckpt_steps"]):
        s.run(int(step) - done); done = int(step)
        tm1, tm2 = s.download()
        out.append("%d: %.2e" % (int(step), max(H.rel_linf(tm1.astype(np.float64), g["ckpt_tm1"][k].astype(np.float64)),
                                                  H.rel_linf(tm2.astype(np.float64), g["ckpt_tm2"][k].astype(np.float64)))))
    s.close()
    print("c5 two-level octree (one rank), %s: %s" % (label, "  ".join(out)))
