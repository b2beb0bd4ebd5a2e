# round 6: kernel timelines for the record -- a rank of 8 alone (loopback), and kernel stats of the gradient basin
O=gpurun_out/r06_traces; mkdir -p $O
export HQ_ALLOW_ENV=1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/rank -- python3 $GRAFT_REPO_ROOT/profiles/tools/rank_alone_trace.py 3 60 c3 > /dev/null 2>&1 )
f=$(find $O/rank -name "*kernel_trace.csv" | head -1); python3 profiles/tools/rank_alone_trace.py --analyse $f | cut -c1-400 > $O/rank_alone_trace.txt; head -30 $O/rank_alone_trace.txt
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/o4g -- python3 $GRAFT_REPO_ROOT/bench.py --workload o4g --no-pmc --no-cpu-baseline --no-parity --repeats 2 --steps 50 > $GRAFT_REPO_ROOT/$O/bench_o4g_under_rocprof.json 2>/dev/null )
head -12 $(find $O/o4g -name "*kernel_stats.csv" | head -1) | cut -c1-100,240-330
rm -rf $O/rank $O/o4g/*/*kernel_trace.csv
