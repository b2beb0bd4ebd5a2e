#!/usr/bin/env python3
"""One recorded run of the REAL reference (oracle/_ref/psolve, all host cores as MPI ranks) on the 64 M-element box of
BASELINE configs[2]: examples/simple's material refined by the reference's own mesher at f = 160 Hz to 512 x 512 x 256
elements, 32 point sources so that no element is quiescent, the solver's own wall clock over steps 50..100
("the reference CPU path timed on the same box's host cores ... memory permitting").  Too slow for every
bench.py run (minutes of meshing, ~90 GB): run once per round, output kept under profiles/.

    python profiles/tools/ref_psolve_64m.py [ranks] > profiles/rNN/ref_psolve_64m.json
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402  (usable_cores)
from oracle import ref_baseline as rb          # noqa: E402

ranks = int(sys.argv[1]) if len(sys.argv) > 1 else bench.usable_cores()
# the reference skips quiescent elements (quake_util.c:49-68) and activity spreads one element per step from the 4 x 4 x 2
# sources (128 elements apart): every element is active after ~111 steps, so the clock is read over steps 150..200
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 201
t0 = time.time()
r = rb.measure_box(ranks, freq=160.0, dt=0.000125, steps=steps, timeout=3400)
r["wall_s_whole_run"] = time.time() - t0
r["command"] = "mpiexec -np %d oracle/_ref/psolve parameters.in  (f = 160 Hz, dt = 1.25e-4, %d steps)" % (ranks, steps)
r["per_core"] = r["value"] / ranks
print(json.dumps(r))
