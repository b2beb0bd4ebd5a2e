# round 6: the 8 M box's 129 planes in 5 chunks (640 units, 1.25 rounds of 512 slots) or 4 (512 units, one round)?
O=gpurun_out/r06_cz; mkdir -p $O
export HQ_ALLOW_ENV=1
for cz in 0 33 0 33 26 43; do
  if [ $cz = 0 ]; then unset HQ_BRICK_CZ; else export HQ_BRICK_CZ=$cz; fi
  timeout 600 python3 bench.py --workload c2 --no-pmc --no-cpu-baseline --repeats 3 --steps 200 > $O/bench_c2_cz$cz.json 2>/dev/null
  python3 - $O/bench_c2_cz$cz.json $cz <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d['config']; r=d['roofline']
print('c2 cz', sys.argv[2], 'ms', round(d['ms_per_step'],4), c.get('ms_per_step_runs'), 'parity', c.get('parity_worst'), 'phase', r.get('phase_us'))
PY
done 2>&1 | tee $O/ab.txt
