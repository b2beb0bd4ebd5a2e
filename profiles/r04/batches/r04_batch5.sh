#!/bin/bash
# round 4, GPU batch 5: two-step micro-benchmark (temporal blocking prototype); in-process proxy with one HW queue per partition
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/b5; mkdir -p $O
( cd /tmp && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/march_twostep $GRAFT_REPO_ROOT/profiles/micro/march_twostep.hip && timeout 600 /tmp/march_twostep > $GRAFT_REPO_ROOT/$O/march_twostep.txt 2>&1 )
cat $O/march_twostep.txt
( cd /tmp && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/march_stencil $GRAFT_REPO_ROOT/profiles/micro/march_stencil.hip && timeout 600 /tmp/march_stencil 2>&1 | head -4 > $GRAFT_REPO_ROOT/$O/march_onestep.txt )
cat $O/march_onestep.txt
for cfg in "default:" "q8:GPU_MAX_HW_QUEUES=8" "q32:GPU_MAX_HW_QUEUES=32"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  ( export $envs HQ_DUMMY=1; timeout 600 python bench.py --steps 100 --warmup 20 --no-pmc --no-cpu-baseline --inproc-parts 8 > $O/inproc8_$name.json 2> $O/inproc8_$name.err )
  echo "inproc8 $name: $(python3 -c "import json;print(json.load(open('$O/inproc8_$name.json'))['ms_per_step'])")"
done
