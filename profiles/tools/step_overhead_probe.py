"""What does one step cost through the C-ABI on a tiny mesh (examples/simple size), call by call?  The stub of
INTEGRATION.md calls hq_set_source + hq_run(1) + hq_download + hq_check_finite per step inside the reference's loop."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hercules_amd as ha
from hercules_amd import host

b = host.Box(16, 16, 8, 62.5, 1e-3, 5.0)
loaded, pattern = b.point_source(500.0, 500.0, 100.0, 0.0, 90.0, 0.0)
rp = b.run_params(loaded=loaded, pattern=pattern, moment=1e15, rise_time=0.02, source_window=32)
F = b.source_table(rp, 0, 2000)
s = b.create_solver()
s.set_source(loaded, F[:1])
n = 1000
for name, fn in (("set_source(1 step)", lambda k: s.set_source(loaded, F[k:k + 1], k)),
                 ("run(1)", lambda k: s.run(1)),
                 ("sync", lambda k: s.sync()),
                 ("download", lambda k: s.download()),
                 ("check_finite", lambda k: s.check_finite()),
                 ("all four", lambda k: (s.set_source(loaded, F[k:k + 1], k), s.run(1), s.download(), s.check_finite()))):
    s.sync()
    t = time.perf_counter()
    for k in range(n):
        fn(k)
    s.sync()
    print("%-20s %8.1f us per call" % (name, (time.perf_counter() - t) / n * 1e6))
