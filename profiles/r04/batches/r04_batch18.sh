#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/b18; mkdir -p $O
HQ_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --workload o3s --steps 30 --warmup 5 > $O/bench_o3s_auto2.json 2> $O/bench_o3s_auto2.err; python3 -c "import json;d=json.load(open('$O/bench_o3s_auto2.json'));print('o3s auto x2 on one GPU', d['ms_per_step'], d['config']['transport'][:120], d['config']['finite'])"; tail -2 $O/bench_o3s_auto2.err
HQ_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 4 --workload o3s --steps 30 --warmup 5 > $O/bench_o3s_auto4.json 2> $O/bench_o3s_auto4.err; python3 -c "import json;d=json.load(open('$O/bench_o3s_auto4.json'));print('o3s auto x4 on one GPU', d['ms_per_step'], d['config']['transport'][:120], d['config']['finite'])"; tail -2 $O/bench_o3s_auto4.err
