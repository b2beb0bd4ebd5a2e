# round 6: 64 x 16 tiles (-DHQ_BK_TY=16: one 1 024-thread workgroup per CU, 16 % instead of 29 % ring rows) against the
# shipped 64 x 8 on one box: the headline, the 8 M box, a rank of 8 alone, the lateral basin
O=gpurun_out/r06_ty16; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 HQ_ALLOW_ENV=1
for v in default ty16 default ty16; do
  if [ $v = default ]; then unset HQ_SOLVER_LIB; else export HQ_SOLVER_LIB=$PWD/profiles/experiments/libhq_solver_ty16.so; fi
  for w in c3 c2; do python bench.py --workload $w --no-cpu-baseline --no-pmc --repeats 3 > $O/bench_${w}_$v.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$O/bench_${w}_$v.json')); print('$w', '$v', round(d['ms_per_step'],4), d['config']['parity_worst'], d['config']['brick_nodes'], d['config']['patches'])"; done
  HQ_TRACE_TIME_STEPS=1000 python3 profiles/tools/rank_alone_trace.py 3 30 c3 2>/dev/null | grep "wall clock" | tr '\n' ' '; echo " <- rank alone $v"
done
for v in default ty16; do
  if [ $v = default ]; then unset HQ_SOLVER_LIB; else export HQ_SOLVER_LIB=$PWD/profiles/experiments/libhq_solver_ty16.so; fi
  python bench.py --workload o4 --no-cpu-baseline --no-pmc --repeats 3 > $O/bench_o4_$v.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$O/bench_o4_$v.json')); print('o4', '$v', round(d['ms_per_step'],4), d['config']['parity_worst'], d['config']['brick_nodes'], d['config']['patches'])"
done
