# round 6: a rank of 8 alone (loopback transport): how much transport latency does the step hide?  The loopback raises every
# flag HQ_LOOPBACK_DELAY_US late (both exchanges of the step), as if the records had a link to cross.
O=gpurun_out/r06_latency; mkdir -p $O
export HQ_TRACE_TIME_STEPS=1000 HQ_ALLOW_ENV=1
for d in 0 5 10 20 30 40 60 80 0; do
  echo "== flags $d us late"
  HQ_LOOPBACK_DELAY_US=$d python3 profiles/tools/rank_alone_trace.py 3 30 c3 2>/dev/null | grep "wall clock"
done 2>&1 | tee $O/rank_alone_latency.txt
