#!/bin/bash
# GPU box: vector-memory path counters of the patch kernel on the 64M box (separate --pmc passes,
# never with tracing).  bash profiles/pmc_vmem.sh <tag>
TAG=${1:-vmem}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline"
run() { local name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$REPO/bench.py" $ARGS > "$OUT/$name.log" 2>&1; echo "$name rc=$?"; }
run a SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES
run b SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES
# (a pass with TA_* counters aborted inside rocprofv3 on this pool and hung until the timeout: not collected)
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for name in "ab":
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("%s/%s/*/*counter_collection.csv" % (out, name)):
        for r in csv.DictReader(open(f)):
            if "hq_k_patch" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in sorted(acc.items()):
        print("%s %-36s %18.0f per launch (%d rows)" % (name, k, v / max(n, 1) , n))
PY
