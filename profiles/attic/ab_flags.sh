#!/bin/bash
# GPU box: A/B of compile-time variants of the solver library (ephemeral rebuilds).
#   FLAGSETS="-DA|-DB -DC|" bash profiles/ab_flags.sh      ('|' separates the sets; empty = as shipped)
IFS='|' read -ra SETS <<< "${FLAGSETS:-|}"
for fl in "${SETS[@]}" ""; do
  [ -z "$fl" ] && [ -n "$did_empty" ] && continue
  [ -z "$fl" ] && did_empty=1
  HQ_EXTRA_FLAGS="$fl" python -c "from hercules_amd import build; build.build_solver(force=True)" > /dev/null
  for w in ${WLS:-c3}; do
    python bench.py --workload $w --steps ${STEPS:-60} --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$fl] $w', round(d['value']/1e9,2), 'G/s', round(d['ms_per_step'],4), 'ms')"
  done
done
