for i in 1 2 3 4 5 6 7 8; do
  HSA_ENABLE_IPC_MODE_LEGACY=0 HQ_BENCH_SHARE_GPU=1 HQ_BENCH_TRANSPORT=ipc python bench.py --gpus 8 --workload c2 --steps 10 --warmup 3 > /tmp/l8_$i.out 2> /tmp/l8_$i.err; rc=$?
  echo "run $i rc $rc $(cut -c150-230 /tmp/l8_$i.out | head -1)"
  if [ $rc -ne 0 ]; then tail -15 /tmp/l8_$i.err | cut -c1-300; fi
done
