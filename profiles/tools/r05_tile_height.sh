# experiment builds of hq_k_brick with 64 x 4 and 64 x 16 tiles (-DHQ_BK_TY) against the shipped 64 x 8, one box
O=gpurun_out/r05_ty; mkdir -p $O
for v in default ty4 ty16 default; do
  if [ $v = default ]; then unset HQ_SOLVER_LIB; else export HQ_SOLVER_LIB=$PWD/profiles/experiments/libhq_solver_$v.so; fi
  python bench.py --no-cpu-baseline --no-pmc > $O/bench_c3_$v.json 2>$O/err_$v.txt; echo $v; cut -c150-260 $O/bench_c3_$v.json; python -c "
import json; d=json.load(open('$O/bench_c3_$v.json')); print(d['config']['brick_nodes'], d['config']['patches'], d['config']['parity_worst'])"
done
