#!/bin/bash
# round 4, GPU batch 3: fused interface update + sharing pack, merged patch launch; trace; proxies; whole GPU suite with durations
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/b3; mkdir -p $O
for cfg in "default:" "split:HQ_PATCH_SPLIT_LAUNCH=1" "nofuse:HQ_NO_FUSED_SHARE=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  rm -rf /tmp/tr_$name
  ( export $envs HQ_DUMMY=1; cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$name -- python3 $GRAFT_REPO_ROOT/profiles/tools/rank_alone_trace.py 3 40 > $GRAFT_REPO_ROOT/$O/trace_$name.log 2>&1 )
  f=$(find /tmp/tr_$name -name "*kernel_trace.csv" | head -1)
  echo "== $name ($envs)" >> $O/rank_alone_trace.txt
  python3 profiles/tools/rank_alone_trace.py --analyse "$f" >> $O/rank_alone_trace.txt 2>&1
done
cat $O/rank_alone_trace.txt
timeout 600 python bench.py --steps 100 --warmup 20 --no-pmc --no-cpu-baseline --inproc-parts 8 > $O/inproc8_c3.json 2> $O/inproc8_c3.err; tail -c 300 $O/inproc8_c3.json
HQ_OVERLAP=1 timeout 600 python bench.py --steps 100 --warmup 20 --no-pmc --no-cpu-baseline --inproc-parts 8 > $O/inproc8_c3_ov.json 2> $O/inproc8_c3_ov.err; tail -c 300 $O/inproc8_c3_ov.json
HQ_BENCH_SHARE_GPU=1 HQ_BENCH_TRANSPORT=ipc timeout 600 python bench.py --gpus 2 --workload c2 --steps 100 --warmup 20 > $O/bench_c2_ipc2.json 2> $O/bench_c2_ipc2.err; python3 -c "import json;d=json.load(open('$O/bench_c2_ipc2.json'));print('c2 ipc x2', d['ms_per_step'], d['config']['transport'])"
timeout 2400 python -m pytest tests -m gpu -x -q --durations=70 > $O/pytest_all.log 2>&1; echo "pytest rc $?" >> $O/pytest_all.log
tail -90 $O/pytest_all.log
