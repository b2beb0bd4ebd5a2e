#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/b12; mkdir -p $O
HQ_TRACE_TIME_STEPS=1000 timeout 300 python3 profiles/tools/rank_alone_trace.py 3 40 2>&1 | grep -E "wall clock|rank 3" | cut -c1-200 | tee $O/rank_alone_wallclock.txt
HQ_TRACE_TIME_STEPS=1000 HQ_BRICK_BY_COMPONENT=0 timeout 300 python3 profiles/tools/rank_alone_trace.py 3 40 2>&1 | grep -E "wall clock" | sed 's/^/118-VGPR form: /' | tee -a $O/rank_alone_wallclock.txt
HQ_TRACE_TIME_STEPS=1000 HQ_LOOPBACK_DELAY_US=20 timeout 300 python3 profiles/tools/rank_alone_trace.py 3 40 2>&1 | grep -E "wall clock" | sed 's/^/flags 20 us late: /' | tee -a $O/rank_alone_wallclock.txt
