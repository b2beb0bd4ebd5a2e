# round 6, review item 2c: the IPC bring-up of 8 ranks sharing one GPU, N times back to back, EVERY failing run's
# stdout and stderr kept (gpurun_out/r06_loop8/): is there a bring-up race, and what does it say when it happens?
N=${1:-100}; O=gpurun_out/r06_loop8; mkdir -p $O
ok=0; bad=0
for i in $(seq 1 $N); do
  HSA_ENABLE_IPC_MODE_LEGACY=0 HQ_BENCH_SHARE_GPU=1 HQ_BENCH_TRANSPORT=ipc python bench.py --gpus 8 --workload m1 --steps 5 --warmup 2 --repeats 1 --no-parity > $O/run_$i.out 2> $O/run_$i.err; rc=$?
  if [ $rc -eq 0 ] && grep -q '^{' $O/run_$i.out; then ok=$((ok+1)); rm -f $O/run_$i.out $O/run_$i.err; else bad=$((bad+1)); echo "run $i rc $rc"; tail -20 $O/run_$i.err | cut -c1-300; fi
done
echo "8 ranks over IPC on one GPU: $ok of $N runs came up and finished, $bad failed (their output is kept in $O/)"
