# round 6, item 1: a rank of 8 alone (loopback transport) -- where can the shell's patches go so that they do not run
# ahead of the brick launch?  Wall clock over 1000 steps x 3 per arrangement, one box, back to back.
#   HQ_BRICK_STREAM=1  bricks on a stream of their own, enqueued BEHIND the patches (no dependency between them)
#   HQ_BRICK_STREAM=2  the same, enqueued AHEAD of the patches
#   HQ_PATCH_LIGHT=1   every patch launch as 256-thread workgroups of the 96-register form (fits beside two brick workgroups)
#   HQ_PATCH_LIGHT=2   only the patches that own no interface node
#   HQ_X_NO_SHELL      (experiment build) no patch launch at all: the floor
O=gpurun_out/r06_shell; mkdir -p $O
export HQ_TRACE_TIME_STEPS=1000 HQ_ALLOW_ENV=1
run() { name=$1; shift; echo "== $name: $*"; env "$@" python3 profiles/tools/rank_alone_trace.py 3 30 c3 2>$O/err_$name.txt | grep -v "^rank" ; }
run default HQ_NOP=1
run bs1 HQ_BRICK_STREAM=1
run bs1_light HQ_BRICK_STREAM=1 HQ_PATCH_LIGHT=1
run bs2_light HQ_BRICK_STREAM=2 HQ_PATCH_LIGHT=1
run bs1_split_light2 HQ_BRICK_STREAM=1 HQ_PATCH_MERGE_ROUNDS=0 HQ_PATCH_LIGHT=2
run bs2_split_light2 HQ_BRICK_STREAM=2 HQ_PATCH_MERGE_ROUNDS=0 HQ_PATCH_LIGHT=2
run light_only HQ_PATCH_LIGHT=1
export HQ_SOLVER_LIB=$PWD/profiles/experiments/libhq_solver_x.so
run noshell HQ_X_NO_SHELL=1
run noshell_118 HQ_X_NO_SHELL=1 HQ_BRICK_BY_COMPONENT=0
unset HQ_SOLVER_LIB
run default_again HQ_NOP=1
# kernel timelines of two arrangements
unset HQ_TRACE_TIME_STEPS
trace() { name=$1; shift; ( export "$@"; cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_$name -- python3 $GRAFT_REPO_ROOT/profiles/tools/rank_alone_trace.py 3 60 c3 > /dev/null 2>&1 ); f=$(find $O/trace_$name -name "*kernel_trace.csv" | head -1); echo "== trace $name"; python3 profiles/tools/rank_alone_trace.py --analyse $f | cut -c1-400 | head -40; rm -rf $O/trace_$name; }
trace bs1_light HQ_BRICK_STREAM=1 HQ_PATCH_LIGHT=1
trace bs2_light HQ_BRICK_STREAM=2 HQ_PATCH_LIGHT=1
