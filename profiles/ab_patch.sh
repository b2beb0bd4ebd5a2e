#!/bin/bash
# GPU box: A/B of patch-kernel build/env variants on the 64M box
run() { echo "== $*"; env "$@" python bench.py --workload ${WL:-c3} --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('  G elem/s %.2f  ms %.3f  frac %.3f  kernel_ms %.3f'%(d['value']/1e9,d['ms_per_step'],d['roofline']['frac'],d['roofline']['kernel_ms']))"; }
HQ_PATCH_PIPE=1 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
run HQ_PATCH_PIPE=0
run HQ_PATCH_PIPE=1
run HQ_PATCH_PIPE=0
run HQ_PATCH_PIPE=1
