#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/b11; mkdir -p $O
for i in 1 2 3; do
  rm -rf /tmp/tr_$i
  ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$i -- python3 $GRAFT_REPO_ROOT/profiles/tools/rank_alone_trace.py 3 60 > $GRAFT_REPO_ROOT/$O/trace_$i.log 2>&1 )
  f=$(find /tmp/tr_$i -name "*kernel_trace.csv" | head -1)
  echo "== run $i" >> $O/rank_alone.txt
  python3 profiles/tools/rank_alone_trace.py --analyse "$f" 2>&1 | head -12 >> $O/rank_alone.txt
done
cat $O/rank_alone.txt
timeout 600 python -m pytest tests/test_gpu_multiprocess.py -q -x -s -k "2-1-0-ipc" 2>&1 | grep -E "IPC receive|passed|failed"
