#!/bin/bash
# round 4, GPU batch 10: interface update with shallower loads; hq_k_brick with the plane sums component by component (100 VGPRs)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/b10; mkdir -p $O
run_set() {
  tag=$1
  timeout 600 python bench.py --steps 100 --warmup 20 --no-pmc --no-cpu-baseline > $O/bench_c3_$tag.json 2> $O/bench_c3_$tag.err; python3 -c "import json;d=json.load(open('$O/bench_c3_$tag.json'));print('$tag c3', d['ms_per_step'], d['roofline']['kernel_ms'])"
  timeout 600 python bench.py --steps 100 --warmup 20 --no-pmc --no-cpu-baseline --workload c2 > $O/bench_c2_$tag.json 2> $O/bench_c2_$tag.err; python3 -c "import json;d=json.load(open('$O/bench_c2_$tag.json'));print('$tag c2', d['ms_per_step'])"
  timeout 600 python bench.py --steps 50 --warmup 10 --no-pmc --no-cpu-baseline --workload o3 > $O/bench_o3_$tag.json 2> $O/bench_o3_$tag.err; python3 -c "import json;d=json.load(open('$O/bench_o3_$tag.json'));print('$tag o3', d['ms_per_step'])"
  for d in 0 20 40; do
    rm -rf /tmp/tr_$tag$d
    ( export HQ_LOOPBACK_DELAY_US=$d; cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag$d -- python3 $GRAFT_REPO_ROOT/profiles/tools/rank_alone_trace.py 3 40 > $GRAFT_REPO_ROOT/$O/trace_$tag$d.log 2>&1 )
    f=$(find /tmp/tr_$tag$d -name "*kernel_trace.csv" | head -1)
    echo "== $tag, flags raised $d us late" >> $O/rank_alone.txt
    python3 profiles/tools/rank_alone_trace.py --analyse "$f" 2>&1 | head -9 >> $O/rank_alone.txt
  done
  timeout 600 python bench.py --steps 100 --warmup 20 --no-pmc --no-cpu-baseline --inproc-parts 8 > $O/inproc8_$tag.json 2> $O/inproc8_$tag.err; python3 -c "import json;d=json.load(open('$O/inproc8_$tag.json'));print('$tag inproc8', d['ms_per_step'])"
}
run_set base
HQ_EXTRA_FLAGS=-DHQ_BK_BY_COMPONENT python -m hercules_amd.build --force > $O/rebuild.log 2>&1; tail -1 $O/rebuild.log
run_set bycomp
timeout 900 python -m pytest tests/test_gpu_brick_variants.py tests/test_gpu_multiprocess.py tests/test_gpu_parity.py -q -x -k "brick or own_processes or c1_against or partition" > $O/pytest_bycomp.log 2>&1; tail -3 $O/pytest_bycomp.log
cat $O/rank_alone.txt
