# libhq_solver_f32.so (hq_real = float): its tests, a bench line on the 64 M box, and the fp64 line beside it on the same box
O=gpurun_out/r05_f32; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests/test_gpu_single_precision.py -m gpu -q --durations=6 2>&1 | tail -40 > $O/pytest_f32.log; cat $O/pytest_f32.log
python bench.py --precision f32 --pmc-dir $O/pmc_f32 > $O/bench_c3_f32.json 2> $O/bench_c3_f32.err; cut -c1-2500 $O/bench_c3_f32.json; tail -3 $O/bench_c3_f32.err
python bench.py --no-cpu-baseline --no-pmc > $O/bench_c3_f64.json 2>/dev/null; cut -c150-260 $O/bench_c3_f64.json
python bench.py --precision f32 --workload c3h --no-cpu-baseline --no-pmc > $O/bench_c3h_f32.json 2>/dev/null; cut -c150-260 $O/bench_c3h_f32.json
python bench.py --precision f32 --workload o4 --no-cpu-baseline --no-pmc > $O/bench_o4_f32.json 2>/dev/null; cut -c150-260 $O/bench_o4_f32.json
