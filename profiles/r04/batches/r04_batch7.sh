#!/bin/bash
# round 4, GPU batch 7: shell variants on the rank-alone trace; new tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/b7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_brick_variants.py -q -x --durations=10 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -12 $O/pytest.log
for cfg in "default:" "merge2:HQ_PATCH_MERGE_ROUNDS=2" "p256:HQ_PATCH_PSPLIT=256 HQ_PATCH_PMERGE=256" "p256m:HQ_PATCH_PSPLIT=256 HQ_PATCH_PMERGE=256 HQ_PATCH_MERGE_ROUNDS=4" "t256:HQ_PATCH_THREADS=256" "pipe6:HQ_PATCH_PIPE=6" "cz64:HQ_BRICK_CZ=64"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  rm -rf /tmp/tr_$name
  ( export $envs HQ_DUMMY=1; cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$name -- python3 $GRAFT_REPO_ROOT/profiles/tools/rank_alone_trace.py 3 40 > $GRAFT_REPO_ROOT/$O/trace_$name.log 2>&1 )
  f=$(find /tmp/tr_$name -name "*kernel_trace.csv" | head -1)
  echo "== $name ($envs)" >> $O/rank_alone_trace.txt
  python3 profiles/tools/rank_alone_trace.py --analyse "$f" 2>&1 | head -14 >> $O/rank_alone_trace.txt
done
cat $O/rank_alone_trace.txt
