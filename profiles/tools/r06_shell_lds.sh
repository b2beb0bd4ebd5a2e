# round 6, item 1 (second batch): the shell's patch launch with its LDS image cut to what the patches stage
# (HQ_PATCH_TIGHT_LDS, default on), 256-thread / 96-register workgroups, and the brick forms -- a rank of 8 alone,
# wall clock over 1000 steps x 3, arrangements interleaved because a box drifts by several percent within minutes
O=gpurun_out/r06_shell; mkdir -p $O
export HQ_TRACE_TIME_STEPS=1000 HQ_ALLOW_ENV=1
run() { name=$1; shift; echo "== $name: $*"; env "$@" python3 profiles/tools/rank_alone_trace.py 3 30 c3 2>$O/err_$name.txt | grep -v "^rank" ; }
run warmup HQ_NOP=1
for rep in 1 2; do
run default HQ_NOP=1
run loose_lds HQ_PATCH_TIGHT_LDS=0
run light HQ_PATCH_LIGHT=1
run threads256 HQ_PATCH_THREADS=256
run bricks118 HQ_BRICK_BY_COMPONENT=0
run pmerge768 HQ_PATCH_PMERGE=768
done
run default HQ_NOP=1
grep -h "patches\|hq patch plan" $O/err_default.txt | head -5
