# round 6: two-stream steps (shell beside bricks) -- which stream should the dispatcher prefer?  (experiment build)
O=gpurun_out/r06_prio; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 HQ_SOLVER_LIB=$PWD/profiles/experiments/libhq_solver_x.so
for w in o3 o4; do
for p in low normal high low; do
  if [ $p = low ]; then unset HQ_X_BRICK_PRIO; else export HQ_X_BRICK_PRIO=$p; fi
  python bench.py --workload $w --no-cpu-baseline --no-pmc --no-parity --repeats 3 > $O/bench_${w}_$p.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$O/bench_${w}_$p.json')); print('$w', '$p', round(d['ms_per_step'],4), d['roofline']['phase_us'])"
done; done
