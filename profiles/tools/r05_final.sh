# the evidence of the final build: GPU suite, default bench line + counters, the driver's command, kernel stats
O=gpurun_out/r05_final; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python -m pytest tests -m gpu -x -q --durations=40 2>&1 | tail -60 > $O/pytest_gpu_final.log
python bench.py --pmc-dir $O/pmc_default > $O/bench_default.json 2> $O/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>> $O/bench_default.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kstats -- python3 $GRAFT_REPO_ROOT/bench.py --no-pmc --no-cpu-baseline --no-parity > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2>/dev/null )
python bench.py --workload c3h --no-cpu-baseline --no-pmc > $O/bench_c3h.json 2>/dev/null
tail -2 $O/smoke.log; cat $O/pytest_gpu_final.log; cat $O/bench_default.json | cut -c1-3000; cat $O/bench_driver_cmd.json | cut -c1-400; head -5 $(find $O/kstats -name "*kernel_stats.csv" | head -1) | cut -c1-200; cut -c1-200 $O/bench_c3h.json
python bench.py --workload o4 --no-cpu-baseline --pmc-dir $O/pmc_o4 > $O/bench_o4.json 2>/dev/null; cut -c1-200 $O/bench_o4.json
python bench.py --workload o3 --no-cpu-baseline --no-pmc > $O/bench_o3.json 2>/dev/null; cut -c150-260 $O/bench_o3.json
