#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root:
#   bash profiles/run_profiles.sh <tag> [bench args...]
# Collects, each in its OWN rocprofv3 run (never --pmc together with tracing):
#   1. --kernel-trace --stats of bench.py               -> kernel durations
#   2. --pmc FETCH_SIZE ; --pmc WRITE_SIZE              -> HBM traffic (MI355X_MICROARCH.md s HBM)
#   3. --pmc SQ_* / TCC_* groups                        -> stalls, LDS conflicts, L2 hit rate
# Raw output lands in gpurun_out/prof_<tag>/, the digest in gpurun_out/prof_<tag>/summary_<tag>.json
set -u
TAG=${1:-r01}; shift || true
ARGS=${@:---steps 10 --warmup 2 --no-cpu-baseline}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { # name, rocprof flags...
    local name=$1; shift
    rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 "$REPO/bench.py" $ARGS > "$OUT/$name.log" 2>&1
    echo "$name rc=$?"
}
run trace --kernel-trace --stats
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run pmc_sq2 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
run pmc_tcc --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
python3 "$REPO/profiles/summarize.py" "$OUT" "$TAG" > "$OUT/summary_$TAG.json"
cat "$OUT/summary_$TAG.json"
