#!/bin/bash
# GPU box: A/B of patch-kernel env variants on the 64M box: bash profiles/ab_patch.sh "A=1 B=2" "A=3" ...
run() { echo "== $*"; env $* timeout 150 python bench.py --workload ${WL:-c3} --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('  G elem/s %.2f  ms %.3f  frac %.3f  kernel_ms %.3f'%(d['value']/1e9,d['ms_per_step'],d['roofline']['frac'],d['roofline']['kernel_ms']))"; }
for rep in 1 2; do for v in "$@"; do run $v; done; done
