# round 6: the brick launches without the barrier bit (hipExtAnyOrderLaunch) on a context that steps alone -- A/B on one box
O=gpurun_out/r06_anyorder; mkdir -p $O
export HQ_ALLOW_ENV=1
for wl in c3 c2 c3h m1; do
  for ao in 0 1 0 1; do
    HQ_BRICK_ANYORDER=$ao timeout 600 python3 bench.py --workload $wl --no-pmc --no-cpu-baseline --repeats 3 --steps 40 > $O/bench_${wl}_ao$ao.json 2> $O/err_${wl}_ao$ao.txt
    python3 - $O/bench_${wl}_ao$ao.json $wl $ao <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    c=d['config']; r=d['roofline']
    print(sys.argv[2], 'anyorder', sys.argv[3], 'ms', round(d['ms_per_step'],4), c.get('ms_per_step_runs'), 'parity', c.get('parity_worst'), 'phase', r.get('phase_us'))
except Exception as e:
    print(sys.argv[2], sys.argv[3], 'FAILED', e)
PY
  done
done 2>&1 | tee $O/ab.txt
# the timeline: do the kernels overlap?
export HQ_BRICK_ANYORDER=1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --workload c3 --no-pmc --no-cpu-baseline --no-parity --repeats 1 --steps 10 --warmup 2 > /dev/null 2>&1 )
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY' | tee $O/timeline_c3_anyorder.txt
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'hq_k_' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=[r for r in rows if 'count_nonfinite' not in r['Kernel_Name']][-12:]
t0=int(rows[0]['Start_Timestamp'])
for r in rows:
    print('%-40s %10.1f %10.1f us' % (r['Kernel_Name'][:40], (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3))
PY
rm -rf $O/trace
