#!/bin/bash
# round 4, GPU batch 4: bricks on their own stream beside the patches; HW-queue count in the in-process proxy; PCIe bytes of a 64M run
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/b4; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_multiprocess.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -q -x -k "own_processes or eight_partitions or small_basin_on_eight or partition" --durations=15 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -25 $O/pytest.log
for cfg in "default:" "nobstream:HQ_BRICK_STREAM=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  rm -rf /tmp/tr_$name
  ( export $envs HQ_DUMMY=1; cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$name -- python3 $GRAFT_REPO_ROOT/profiles/tools/rank_alone_trace.py 3 40 > $GRAFT_REPO_ROOT/$O/trace_$name.log 2>&1 )
  f=$(find /tmp/tr_$name -name "*kernel_trace.csv" | head -1)
  echo "== $name ($envs)" >> $O/rank_alone_trace.txt
  python3 profiles/tools/rank_alone_trace.py --analyse "$f" >> $O/rank_alone_trace.txt 2>&1
done
cat $O/rank_alone_trace.txt
for cfg in "plain:" "ov:HQ_OVERLAP=1" "q16:GPU_MAX_HW_QUEUES=16" "ov_q16:HQ_OVERLAP=1 GPU_MAX_HW_QUEUES=16" "ov_q16_nobs:HQ_OVERLAP=1 GPU_MAX_HW_QUEUES=16 HQ_BRICK_STREAM=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  ( export $envs HQ_DUMMY=1; timeout 600 python bench.py --steps 100 --warmup 20 --no-pmc --no-cpu-baseline --inproc-parts 8 > $O/inproc8_$name.json 2> $O/inproc8_$name.err )
  echo "inproc8 $name: $(python3 -c "import json;print(json.load(open('$O/inproc8_$name.json'))['ms_per_step'])")"
done
mkdir -p /tmp/mini && timeout 900 examples/hq_psolve_mini 512 512 256 1.953125 9e-5 200 400 /tmp/mini > $O/psolve_mini_c3.txt 2>&1; tail -4 $O/psolve_mini_c3.txt
timeout 600 python bench.py --steps 100 --warmup 20 --no-pmc --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; python3 -c "import json;d=json.load(open('$O/bench_c3.json'));print('c3', d['ms_per_step'], d['roofline']['kernel_ms'])"
