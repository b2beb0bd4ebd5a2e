O=gpurun_out/r05_ragged3; mkdir -p $O
for pipe in 0 6 4; do
  HQ_PATCH_PIPE=$pipe python bench.py --workload o4 --no-cpu-baseline --no-pmc --no-parity > $O/bench_o4_pipe$pipe.json 2>/dev/null
  echo pipe $pipe; cut -c150-260 $O/bench_o4_pipe$pipe.json
done
HQ_BRICK_RAGGED=0 python bench.py --workload o4 --no-cpu-baseline --no-pmc --no-parity > $O/bench_o4_ragged0.json 2>/dev/null; echo ragged0; cut -c150-260 $O/bench_o4_ragged0.json
HQ_PATCH_PIPE=6 HQ_BRICK_RAGGED_MINFILL=64 python bench.py --workload o4 --no-cpu-baseline --no-pmc --no-parity > $O/bench_o4_pipe6_mf64.json 2>/dev/null; echo pipe6 mf64; cut -c150-260 $O/bench_o4_pipe6_mf64.json
HQ_PATCH_PIPE=6 python bench.py --workload o4s --no-cpu-baseline --no-pmc --no-parity > $O/bench_o4s_pipe6.json 2>/dev/null; echo o4s pipe6; cut -c150-260 $O/bench_o4s_pipe6.json
python bench.py --workload o4s --no-cpu-baseline --no-pmc --no-parity > $O/bench_o4s.json 2>/dev/null; echo o4s; cut -c150-260 $O/bench_o4s.json
