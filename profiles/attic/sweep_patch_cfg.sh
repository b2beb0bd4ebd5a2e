#!/bin/bash
# GPU box: sweep the patch-kernel tuning knobs on one workload (WL=c3|c2).
run() { echo "== $*"; env "$@" python bench.py --workload ${WL:-c3} --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('  G elem/s %.2f  ms %.3f  frac %.3f  patches %d pairs %d'%(d['value']/1e9,d['ms_per_step'],d['roofline']['frac'],d['config']['patches'],d['config']['patch_elements']))"; }
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
run HQ_PATCH_THREADS=512
run HQ_PATCH_THREADS=256 HQ_PATCH_PMAX=384 HQ_PATCH_PMERGE=128 HQ_PATCH_NLMAX=640
run HQ_PATCH_THREADS=512 HQ_PATCH_PMAX=384 HQ_PATCH_PMERGE=128 HQ_PATCH_NLMAX=640
run HQ_PATCH_THREADS=256 HQ_PATCH_PMAX=192 HQ_PATCH_PMERGE=64 HQ_PATCH_NLMAX=400
run HQ_PATCH_THREADS=128 HQ_PATCH_PMAX=192 HQ_PATCH_PMERGE=64 HQ_PATCH_NLMAX=400
run HQ_PATCH_THREADS=512 HQ_PATCH_PMAX=1536 HQ_PATCH_PMERGE=1024 HQ_PATCH_NLMAX=2048
