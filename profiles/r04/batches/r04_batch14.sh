#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/b14; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multiprocess.py tests/test_gpu_brick_variants.py -q -x --durations=8 > $O/pytest.log 2>&1; tail -14 $O/pytest.log
