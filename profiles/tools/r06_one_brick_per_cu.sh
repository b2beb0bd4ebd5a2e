# round 6: does ONE brick workgroup per CU (half the residency) still stream at the kernel's rate?  If so a partition could
# leave half of every CU to the shell's patches and the chain.  Experiment build: HQ_X_BRICK_LDS_PAD bytes of unused dynamic LDS.
O=gpurun_out/r06_one_per_cu; mkdir -p $O
export HQ_ALLOW_ENV=1 HQ_SOLVER_LIB=$PWD/profiles/experiments/libhq_solver_x.so
for wl in c2 c3; do
  for pad in 0 53248 0 53248; do
    HQ_X_BRICK_LDS_PAD=$pad timeout 600 python3 bench.py --workload $wl --no-pmc --no-cpu-baseline --repeats 3 --steps 40 > $O/bench_${wl}_pad$pad.json 2> $O/err_${wl}_pad$pad.txt
    python3 - $O/bench_${wl}_pad$pad.json $wl $pad <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d['config']; r=d['roofline']
    print(sys.argv[2], 'pad', sys.argv[3], 'ms', round(d['ms_per_step'],4), c.get('ms_per_step_runs'), 'parity', c.get('parity_worst'), 'phase', r.get('phase_us'))
except Exception as e:
    print(sys.argv[2], sys.argv[3], 'FAILED', e)
PY
  done
done 2>&1 | tee $O/ab.txt
# a rank of 8 alone: bricks one per CU on their own stream, the shell and the chain beside them
export HQ_TRACE_TIME_STEPS=1000
run() { name=$1; shift; echo "== $name: $*"; env "$@" python3 profiles/tools/rank_alone_trace.py 3 30 c3 2>$O/err_$name.txt | grep -v "^rank" ; }
{ run default HQ_NOP=1
  run pad_only HQ_X_BRICK_LDS_PAD=53248
  run bs1_pad HQ_BRICK_STREAM=1 HQ_X_BRICK_LDS_PAD=53248
  run bs1_pad_split HQ_BRICK_STREAM=1 HQ_PATCH_MERGE_ROUNDS=0 HQ_X_BRICK_LDS_PAD=53248
  run default_again HQ_NOP=1; } 2>&1 | tee $O/rank_alone.txt
