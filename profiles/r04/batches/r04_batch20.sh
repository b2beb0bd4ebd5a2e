#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/b20; mkdir -p $O
HQ_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --workload c2 --steps 50 --warmup 10 > $O/bench_c2_auto2.json 2> $O/bench_c2_auto2.err; python3 -c "import json;d=json.load(open('$O/bench_c2_auto2.json'));print('c2 auto x2', d['ms_per_step'], d['config']['transport'][:110], d['config']['finite'])"
HQ_BENCH_SHARE_GPU=1 HQ_BENCH_TRANSPORT=host timeout 600 python bench.py --gpus 2 --workload c2 --steps 50 --warmup 10 > $O/bench_c2_host2.json 2> $O/bench_c2_host2.err; python3 -c "import json;d=json.load(open('$O/bench_c2_host2.json'));print('c2 host x2', d['ms_per_step'], d['config']['transport'][:110], d['config']['finite'])"
timeout 300 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; python3 -c "import json;d=json.load(open('$O/bench_c3.json'));print('c3', d['ms_per_step'])"
