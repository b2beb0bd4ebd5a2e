#!/bin/bash
# Experiment builds of hq_k_brick_het timed on one box: `het_variants.sh "<flags 1>" "<flags 2>" ...` ("" = as shipped).
# Results of -DHQ_BH_ABL=n builds are wrong by construction (no parity check); the shipped library is rebuilt at the end.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03
out=gpurun_out/r03/het_variants.txt
[ -n "$HQ_VARIANT_APPEND" ] || : > $out
wl=${HQ_VARIANT_WORKLOAD:-c3h}
for flags in "$@"; do
  HQ_EXTRA_FLAGS="$flags" python -m hercules_amd.build --force > /dev/null 2>&1 || { echo "build failed: $flags" >> $out; continue; }
  python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline 2> /dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('%-40s %s ms_per_step %.4f kernel_ms %.4f' % ('$flags' or '(shipped)', '$wl', j['ms_per_step'], j['roofline']['kernel_ms']))
" >> $out
done
python -m hercules_amd.build --force > /dev/null 2>&1
cat $out
