#!/bin/bash
# single-GPU shell variants: smaller patches / 256-thread workgroups for contexts without a transport
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/b13; mkdir -p $O
for cfg in "base:" "p256:HQ_PATCH_PSPLIT=256 HQ_PATCH_PMERGE=256" "p256t:HQ_PATCH_PSPLIT=256 HQ_PATCH_PMERGE=256 HQ_PATCH_THREADS=256" "p256tn:HQ_PATCH_PSPLIT=256 HQ_PATCH_PMERGE=256 HQ_PATCH_THREADS=256 HQ_PATCH_NLMAX=640 HQ_PATCH_PMAX=256" "base2:"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  for wl in c3 c2 o3; do
    ( export $envs HQ_DUMMY=1; timeout 600 python bench.py --steps 100 --warmup 20 --no-pmc --no-cpu-baseline --workload $wl > $O/bench_${wl}_$name.json 2> $O/bench_${wl}_$name.err )
    echo "$name $wl: $(python3 -c "import json;d=json.load(open('$O/bench_${wl}_$name.json'));print(d['ms_per_step'], d['config']['patches'])")"
  done
done
