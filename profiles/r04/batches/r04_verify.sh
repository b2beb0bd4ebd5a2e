#!/bin/bash
# final verification of the tree as committed: whole GPU suite, smoke, the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/verify; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest_all.log 2>&1; echo "pytest rc $?" >> $O/pytest_all.log
tail -20 $O/pytest_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; python3 -c "import json;d=json.load(open('$O/bench_driver_cmd.json'));print('c3 driver cmd', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['counter_frac'])"
