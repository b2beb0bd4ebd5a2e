#!/bin/bash
# GPU box: rebuild with -DHQ_PATCH_PROFILING (ephemeral copy) and print the mean shader cycles
# each phase of the patch kernel takes per workgroup (per patch) on the 64M box.
#   [STAMP_TIDS="0 512 960"] bash profiles/stamp_phases.sh [HQ_PATCH_PIPE values...]   (default: 0)
# STAMP_TIDS: the thread of hq_k_patch_pers whose clock is recorded (wave 0 draws the tickets,
# waves 12-15 have no element of a 729-element patch).
for st in ${STAMP_TIDS:-0}; do
HQ_EXTRA_FLAGS="-DHQ_PATCH_PROFILING -DHQ_STAMP_TID=$st" python -c "from hercules_amd import build; build.build_solver(force=True)" > /dev/null
for pipe in ${@:-0}; do
  echo "== HQ_PATCH_PIPE=$pipe stamps of thread $st"
  HQ_PATCH_PIPE=$pipe HQ_PATCH_DIAG=6 python bench.py --workload ${WL:-c3} --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep -B1 -A8 "hq patch stamps"
done
done
