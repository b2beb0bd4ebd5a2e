# round 6: bench.py on 8 ranks sharing this GPU over IPC WITH its parity windows, N times back to back (the bring-up loop ran
# with --no-parity and so could not see what the suite then did: one run in a few hundred with parity_worst 0.37 -- traced to
# a null-stream hipMemset in hq_upload that was not ordered with the context's streams)
N=${1:-30}; O=gpurun_out/r06_loop8p; mkdir -p $O
ok=0; bad=0
for i in $(seq 1 $N); do
  HSA_ENABLE_IPC_MODE_LEGACY=0 HQ_BENCH_SHARE_GPU=1 HQ_BENCH_TRANSPORT=ipc python bench.py --gpus 8 --workload c2 --steps 10 --warmup 3 --repeats 1 > $O/run_$i.out 2> $O/run_$i.err; rc=$?
  if [ $rc -eq 0 ] && grep -q '^{' $O/run_$i.out; then ok=$((ok+1)); rm -f $O/run_$i.out $O/run_$i.err; else bad=$((bad+1)); echo "run $i rc $rc"; grep -h "PARITY\|error\|Error" $O/run_$i.err | head -5 | cut -c1-300; fi
done
echo "8 ranks of c2 over IPC on one GPU, parity windows on: $ok of $N runs passed, $bad failed (kept in $O/)"
