#!/bin/bash
# Experiment builds of the brick kernels timed on one box over several workloads:
#   brick_variants.sh "<workloads>" "<flags 1>" "<flags 2>" ...      ("" = as shipped)
# The shipped library is rebuilt at the end.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03
out=gpurun_out/r03/brick_variants.txt
[ -n "$HQ_VARIANT_APPEND" ] || : > $out
wls="$1"; shift
for flags in "$@"; do
  HQ_EXTRA_FLAGS="$flags" python -m hercules_amd.build --force > /dev/null 2>&1 || { echo "build failed: $flags" >> $out; continue; }
  for wl in $wls; do
    python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline 2> /dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('%-32s %-4s ms_per_step %.4f kernel_ms %.4f  %.2f G/s' % ('$flags' or '(shipped)', '$wl', j['ms_per_step'], j['roofline']['kernel_ms'], j['value'] / 1e9))
" >> $out
  done
done
python -m hercules_amd.build --force > /dev/null 2>&1
cat $out
