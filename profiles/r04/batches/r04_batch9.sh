#!/bin/bash
# round 4, GPU batch 9: how much transport latency does a rank's step hide (loopback with delayed flags); bench transport auto-trial with ranks sharing a GPU
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/b9; mkdir -p $O
for d in 0 10 20 40 80; do
  rm -rf /tmp/tr_d$d
  ( export HQ_LOOPBACK_DELAY_US=$d; cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_d$d -- python3 $GRAFT_REPO_ROOT/profiles/tools/rank_alone_trace.py 3 40 > $GRAFT_REPO_ROOT/$O/trace_d$d.log 2>&1 )
  f=$(find /tmp/tr_d$d -name "*kernel_trace.csv" | head -1)
  echo "== flags raised $d us late (HQ_LOOPBACK_DELAY_US=$d), both exchanges of a step" >> $O/rank_alone_latency.txt
  python3 profiles/tools/rank_alone_trace.py --analyse "$f" 2>&1 | head -12 >> $O/rank_alone_latency.txt
done
cat $O/rank_alone_latency.txt
HQ_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --workload c2 --steps 100 --warmup 20 > $O/bench_c2_auto2.json 2> $O/bench_c2_auto2.err; python3 -c "import json;d=json.load(open('$O/bench_c2_auto2.json'));print('c2 auto x2', d['ms_per_step'], d['config']['transport'], d['config']['transport_trials_ms_per_step'])"; tail -3 $O/bench_c2_auto2.err
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -x -k "eight_partitions and m1" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
