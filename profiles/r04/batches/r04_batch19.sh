#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/b19; mkdir -p $O
for cfg in "q16:" "q32_ov:GPU_MAX_HW_QUEUES=32 HQ_OVERLAP=1" "q24_ov:GPU_MAX_HW_QUEUES=24 HQ_OVERLAP=1" "q16_light:HQ_BRICK_BY_COMPONENT=1" "q32_ov_light0:GPU_MAX_HW_QUEUES=32 HQ_OVERLAP=1 HQ_BRICK_BY_COMPONENT=0" "q16_b:"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  ( export $envs HQ_DUMMY=1; timeout 400 python bench.py --steps 100 --warmup 20 --no-pmc --no-cpu-baseline --inproc-parts 8 > $O/inproc8_$name.json 2> $O/inproc8_$name.err )
  echo "inproc8 $name: $(python3 -c "import json;print(json.load(open('$O/inproc8_$name.json'))['ms_per_step'])" 2>&1 | tail -1)"
done
