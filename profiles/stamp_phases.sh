#!/bin/bash
# GPU box: rebuild with -DHQ_PATCH_PROFILING (ephemeral copy) and print the mean shader cycles
# each phase of hq_k_patch_step takes per workgroup on the 64M box.
HQ_EXTRA_FLAGS=-DHQ_PATCH_PROFILING python -c "from hercules_amd import build; build.build_solver(force=True)"
HQ_PATCH_DIAG=6 python bench.py --workload ${WL:-c3} --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep -A9 "hq patch stamps"
HQ_PATCH_DIAG=6 HQ_PATCH_THREADS=256 python bench.py --workload ${WL:-c3} --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep -A9 "hq patch stamps"
