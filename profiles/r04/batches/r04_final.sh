#!/bin/bash
# round 4, final GPU batch: whole GPU suite (durations), smoke, default bench with counters, kernel stats, other workloads
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/fin; mkdir -p $O $O/pmc $O/pmc_c3h
( cd /tmp && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -w -o /tmp/stream_bw $GRAFT_REPO_ROOT/profiles/micro/stream_bw.hip && timeout 300 /tmp/stream_bw > $GRAFT_REPO_ROOT/$O/stream_bw.txt 2>&1 ); tail -5 $O/stream_bw.txt
timeout 2400 python -m pytest tests -m gpu -x -q --durations=25 > $O/pytest_all.log 2>&1; echo "pytest rc $?" >> $O/pytest_all.log
tail -34 $O/pytest_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py --pmc-dir $O/pmc > $O/bench_default.json 2> $O/bench_default.err; python3 -c "import json;d=json.load(open('$O/bench_default.json'));print('c3 default', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['counter_frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; python3 -c "import json;d=json.load(open('$O/bench_driver_cmd.json'));print('c3 driver cmd', d['ms_per_step'], d['roofline']['kernel_ms'])"
( cd /tmp; rm -rf /tmp/st_c3; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_c3 -- python3 $GRAFT_REPO_ROOT/bench.py --no-pmc --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/stats_run.log 2>&1 ); f=$(find /tmp/st_c3 -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_c3_default_bench.csv; head -6 $O/kernel_stats_c3_default_bench.csv
timeout 900 python bench.py --workload c3h --no-cpu-baseline --pmc-dir $O/pmc_c3h > $O/bench_c3h.json 2> $O/bench_c3h.err; python3 -c "import json;d=json.load(open('$O/bench_c3h.json'));print('c3h', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['counter_frac'], d['roofline']['traffic'])"
for wl in c2 c2h o3 o3s m1; do timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-pmc > $O/bench_$wl.json 2> $O/bench_$wl.err; python3 -c "import json;d=json.load(open('$O/bench_$wl.json'));print('$wl', d['ms_per_step'], d['value'])"; done
timeout 600 python bench.py --steps 100 --warmup 20 --no-pmc --no-cpu-baseline --inproc-parts 8 > $O/inproc8_c3.json 2> $O/inproc8_c3.err; tail -c 250 $O/inproc8_c3.json
timeout 900 python bench.py --steps 50 --warmup 10 --no-pmc --no-cpu-baseline --inproc-parts 8 --workload o3 > $O/inproc8_o3.json 2> $O/inproc8_o3.err; tail -c 250 $O/inproc8_o3.json
rm -rf /tmp/tr_fin; ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_fin -- python3 $GRAFT_REPO_ROOT/profiles/tools/rank_alone_trace.py 3 40 > $GRAFT_REPO_ROOT/$O/trace_fin.log 2>&1 ); f=$(find /tmp/tr_fin -name "*kernel_trace.csv" | head -1); python3 profiles/tools/rank_alone_trace.py --analyse "$f" > $O/rank_alone_trace_final.txt 2>&1; head -16 $O/rank_alone_trace_final.txt
