# ragged brick units, second pass: only the neighbours of owned nodes are loaded; kernel trace of o4
O=gpurun_out/r05_ragged2; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "small_lateral_basin" 2>&1 | tail -3
for v in 0 1; do
  HQ_BRICK_RAGGED=$v python bench.py --workload o4 --no-cpu-baseline --no-pmc --no-parity > $O/bench_o4_ragged$v.json 2>/dev/null
  cut -c150-260 $O/bench_o4_ragged$v.json
done
for mf in 32 64 96; do
  HQ_BRICK_RAGGED_MINFILL=$mf python bench.py --workload o4 --no-cpu-baseline --no-pmc --no-parity > $O/bench_o4_minfill$mf.json 2>/dev/null
  cut -c150-260 $O/bench_o4_minfill$mf.json
done
HQ_BRICK_RAGGED=0 python bench.py --workload o4 --no-cpu-baseline --no-pmc --no-parity > $O/bench_o4_ragged0_again.json 2>/dev/null; cut -c150-260 $O/bench_o4_ragged0_again.json
python bench.py --workload o4 --no-cpu-baseline --no-pmc --no-parity > $O/bench_o4_ragged1_again.json 2>/dev/null; cut -c150-260 $O/bench_o4_ragged1_again.json
HQ_BRICK_STREAM=0 python bench.py --workload o4 --no-cpu-baseline --no-pmc --no-parity > $O/bench_o4_onestream.json 2>/dev/null; cut -c150-260 $O/bench_o4_onestream.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_o4 -- python3 $GRAFT_REPO_ROOT/bench.py --workload o4 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-parity > $GRAFT_REPO_ROOT/$O/bench_o4_under_rocprof.json 2>/dev/null )
f=$(find $O/trace_o4 -name "*kernel_stats.csv" | head -1); cut -c1-150 $f | head -14
t=$(find $O/trace_o4 -name "*kernel_trace.csv" | head -1); python3 - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 3 steps: print kernels with start offsets
names = [r["Kernel_Name"][:40] for r in rows]
last = rows[-40:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print("%-42s q%-3s start %9.1f us  dur %8.1f us" % (r["Kernel_Name"][:42], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
rm -f $t   # (the trace itself is large; the stats and the excerpt above are kept)
