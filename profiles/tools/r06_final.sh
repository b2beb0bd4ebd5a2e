# round 6: the evidence of the final build -- default bench line + counters, the driver's command, kernel stats, the
# other workloads, the single-GPU proxies of an 8-GPU run, the driver's 8-rank form with the ranks sharing this GPU
O=gpurun_out/r06_final; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python bench.py --pmc-dir $O/pmc_default > $O/bench_default.json 2> $O/bench_default.err; cut -c1-330 $O/bench_default.json
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>> $O/bench_default.err; cut -c150-330 $O/bench_driver_cmd.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kstats -- python3 $GRAFT_REPO_ROOT/bench.py --no-pmc --no-cpu-baseline --no-parity > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2>/dev/null )
head -6 $(find $O/kstats -name "*kernel_stats.csv" | head -1) | cut -c1-200
for w in c3h c2 o3; do python bench.py --workload $w --no-cpu-baseline --no-pmc > $O/bench_$w.json 2>/dev/null; echo $w; cut -c150-300 $O/bench_$w.json; done
python bench.py --workload o4 --no-cpu-baseline --pmc-dir $O/pmc_o4 > $O/bench_o4.json 2>/dev/null; echo o4; cut -c150-300 $O/bench_o4.json
# the driver's N = 8 form on the headline workload, the eight ranks sharing this box's one GPU (time-slicing: a correctness
# and plumbing vehicle, not a performance figure)
HQ_BENCH_SHARE_GPU=1 python bench.py --gpus 8 --steps 20 --warmup 5 --repeats 3 > $O/bench_c3_8ranks_one_gpu.json 2> $O/bench_c3_8ranks_one_gpu.err; echo "8 ranks:"; cut -c1-400 $O/bench_c3_8ranks_one_gpu.json; tail -3 $O/bench_c3_8ranks_one_gpu.err
# proxies
export HQ_ALLOW_ENV=1
HQ_TRACE_TIME_STEPS=1000 python3 profiles/tools/rank_alone_trace.py 3 30 c3 2>/dev/null | grep "wall clock" > $O/rank_alone_wallclock.txt; cat $O/rank_alone_wallclock.txt
python bench.py --inproc-parts 8 --workload c3 > $O/inproc8_c3.json 2>/dev/null; cut -c1-200 $O/inproc8_c3.json
# o4g: how full must a plane of a ragged per-element tile be?
for mf in 64 256; do HQ_BRICK_RAGGED_MINFILL=$mf python bench.py --workload o4g --no-cpu-baseline --no-pmc --no-parity --repeats 2 > $O/bench_o4g_minfill$mf.json 2>/dev/null; echo "o4g minfill $mf"; python3 -c "
import json; d=json.load(open('$O/bench_o4g_minfill$mf.json')); print(d['ms_per_step'], d['config']['brick_nodes'], d['config']['patches'], d['roofline']['phase_us'])"; done
