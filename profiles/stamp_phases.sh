#!/bin/bash
# GPU box: rebuild with -DHQ_PATCH_PROFILING (ephemeral copy) and print the mean shader cycles
# each phase of the patch kernel takes per workgroup (per patch) on the 64M box.
#   bash profiles/stamp_phases.sh [HQ_PATCH_PIPE values...]   (default: 0)
HQ_EXTRA_FLAGS=-DHQ_PATCH_PROFILING python -c "from hercules_amd import build; build.build_solver(force=True)"
for pipe in ${@:-0}; do
  echo "== HQ_PATCH_PIPE=$pipe"
  HQ_PATCH_PIPE=$pipe HQ_PATCH_DIAG=6 python bench.py --workload ${WL:-c3} --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep -B2 -A10 "hq patch stamps"
done
