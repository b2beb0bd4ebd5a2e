#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/b17; mkdir -p $O
HQ_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 8 --workload c2 --steps 30 --warmup 5 > $O/bench_c2_auto8.json 2> $O/bench_c2_auto8.err; python3 -c "import json;d=json.load(open('$O/bench_c2_auto8.json'));print('c2 auto x8 on one GPU', d['ms_per_step'], d['config']['transport'][:200], d['config']['transport_trials_ms_per_step'], d['config']['finite'])"; tail -3 $O/bench_c2_auto8.err
