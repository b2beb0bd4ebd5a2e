#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/b16; mkdir -p $O
for cfg in "base:" "p256tn:HQ_PATCH_PSPLIT=256 HQ_PATCH_PMERGE=256 HQ_PATCH_THREADS=256 HQ_PATCH_NLMAX=640 HQ_PATCH_PMAX=256 HQ_PATCH_MERGE_ROUNDS=8" "p384:HQ_PATCH_PSPLIT=384 HQ_PATCH_PMERGE=384 HQ_PATCH_MERGE_ROUNDS=8" "nl768:HQ_PATCH_NLMAX=768 HQ_PATCH_MERGE_ROUNDS=8"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  ( export $envs HQ_DUMMY=1 HQ_TRACE_TIME_STEPS=1000; timeout 300 python3 profiles/tools/rank_alone_trace.py 3 40 2>&1 | grep -E "wall clock|npatches" | cut -c1-160 | sed "s/^/$name: /" ) | tee -a $O/rank_shell_variants.txt
done
