O=gpurun_out/r05_f32b; mkdir -p $O
python -m pytest tests/test_gpu_single_precision.py -m gpu -q 2>&1 | tail -3
python bench.py --precision f32 --no-cpu-baseline --pmc-dir $O/pmc_f32 > $O/bench_c3_f32.json 2>/dev/null; cut -c150-260 $O/bench_c3_f32.json
python bench.py --no-cpu-baseline --no-pmc > $O/bench_c3_f64.json 2>/dev/null; cut -c150-260 $O/bench_c3_f64.json
python bench.py --precision f32 --workload c3h --no-cpu-baseline --no-pmc > $O/bench_c3h_f32.json 2>/dev/null; cut -c150-260 $O/bench_c3h_f32.json
python bench.py --precision f32 --workload o4 --no-cpu-baseline --no-pmc > $O/bench_o4_f32.json 2>/dev/null; cut -c150-260 $O/bench_o4_f32.json
python bench.py --precision f32 --workload c2 --no-cpu-baseline --no-pmc > $O/bench_c2_f32.json 2>/dev/null; cut -c150-260 $O/bench_c2_f32.json
