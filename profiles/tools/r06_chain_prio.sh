# round 6: does the exchange chain's stream priority cost the brick launch beside it?  (experiment build, rank of 8 alone)
O=gpurun_out/r06_chain; mkdir -p $O
export HQ_TRACE_TIME_STEPS=1000 HQ_ALLOW_ENV=1 HQ_SOLVER_LIB=$PWD/profiles/experiments/libhq_solver_x.so
run() { name=$1; shift; echo "== $name: $*"; env "$@" python3 profiles/tools/rank_alone_trace.py 3 30 c3 2>$O/err_$name.txt | grep -v "^rank" ; }
run warm HQ_NOP=1
for rep in 1 2; do
run high HQ_NOP=1
run normal HQ_X_CHAIN_PRIO=normal
run low HQ_X_CHAIN_PRIO=low
done
