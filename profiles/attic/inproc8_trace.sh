#!/bin/bash
# GPU box: kernel trace of the 64M box stepped as 8 in-process partitions on one GPU
# (what one step of the partitioned path launches; timings are serialized on one device).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_inproc8
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --workload ${WL:-c3} --inproc-parts 8 --steps 10 --warmup 2 > "$OUT/trace.log" 2>&1
echo rc=$?
tail -1 "$OUT/trace.log" | cut -c1-300
f=$(ls "$OUT"/trace/*/*_kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print("%-60s calls %6s  total %9.3f ms  avg %9.1f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
