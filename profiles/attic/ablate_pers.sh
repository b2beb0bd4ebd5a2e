#!/bin/bash
# ROUND-1 RECORD: the -DHQ_PERS_DIAG=n ablation bodies this script builds were taken out of the shipping
# translation unit in round 2 (their results are in DESIGN.md s7 and profiles/r01/ablate_pers_v5.txt); the script is kept
# as the record of how those numbers were taken and no longer builds anything different from the shipped kernel.
# GPU box: ablations of hq_k_patch_pers on the 64M box (ephemeral rebuilds with -DHQ_PERS_DIAG=n;
# results are WRONG by construction): 1 = no element section, 2 = no node loads, 3 = no update/stores;
# 7 = halo rows read from behind the owned rows (dense; results wrong): what do the scattered halo lines cost?
# 8 = no descriptor load in the loop (the next descriptor is made up from the previous one; results wrong)
# 9 = 1 + 2 + 3 together: the skeleton of an iteration; 10 = 9 without the end-of-iteration barrier; 11 = 9 without both barriers
# 12 = 9 with a fixed slot order instead of the ticket atomic; 13 = 9 without the source look-up; 14 = both
# 15 = lane-linear local ids in gather and atomics (no LDS bank conflicts; results wrong)
# 4 = probe (results right): two more 8-byte loads per thread per patch (is the load path the limit?).
for d in ${DIAGS:-0 1 2 3}; do
  HQ_EXTRA_FLAGS=-DHQ_PERS_DIAG=$d python -c "from hercules_amd import build; build.build_solver(force=True)" > /dev/null
  echo "== HQ_PERS_DIAG=$d"
  python bench.py --workload ${WL:-c3} --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('  ms %.3f  kernel_ms %.3f'%(d['ms_per_step'],d['roofline']['kernel_ms']))"
done
