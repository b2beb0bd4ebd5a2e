#!/bin/bash
# round 4, GPU batch 1: IPC transport + brick<true> tests, rank-alone trace over the loopback transport, bench sanity
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/b1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multiprocess.py tests/test_gpu_brick_variants.py -x -q --durations=20 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
for cfg in "default:" "cz16:HQ_BRICK_CZ=16" ; do
  name=${cfg%%:*}; envs=${cfg#*:}
  rm -rf /tmp/tr_$name
  ( export $envs HQ_DUMMY=1; cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$name -- python3 $GRAFT_REPO_ROOT/profiles/tools/rank_alone_trace.py 3 40 > $GRAFT_REPO_ROOT/$O/trace_$name.log 2>&1 )
  f=$(find /tmp/tr_$name -name "*kernel_trace.csv" | head -1)
  echo "== $name ($envs)" >> $O/rank_alone_trace.txt
  python3 profiles/tools/rank_alone_trace.py --analyse "$f" >> $O/rank_alone_trace.txt 2>&1
done
cat $O/rank_alone_trace.txt
timeout 600 python bench.py --steps 100 --warmup 20 --no-pmc --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 600 $O/bench_c3.json
HQ_BENCH_SHARE_GPU=1 HQ_BENCH_TRANSPORT=ipc timeout 600 python bench.py --gpus 2 --workload c2 --steps 100 --warmup 20 > $O/bench_c2_ipc2.json 2> $O/bench_c2_ipc2.err; tail -c 400 $O/bench_c2_ipc2.json; tail -3 $O/bench_c2_ipc2.err
HQ_BENCH_SHARE_GPU=1 HQ_BENCH_TRANSPORT=host timeout 600 python bench.py --gpus 2 --workload c2 --steps 100 --warmup 20 > $O/bench_c2_host2.json 2> $O/bench_c2_host2.err; tail -c 400 $O/bench_c2_host2.json
