set -u
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/r02/pmc_ctr_v14
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; timeout -k 5 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$REPO/bench.py" --pmc-child --workload c3 --steps 3 --warmup 1 > "$OUT/$name.log" 2>&1 < /dev/null; echo "$name rc=$?"; }
run a SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run b TCC_HIT_sum TCC_MISS_sum
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for name in "ab":
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("%s/%s/*/*counter_collection.csv" % (out, name)):
        for r in csv.DictReader(open(f)):
            if "hq_k_patch" in r["Kernel_Name"]:
                k = (r["Kernel_Name"][:30], r["Counter_Name"])
                acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for k, (v, n) in sorted(acc.items()):
        print("%s %-32s %-24s %18.0f per dispatch (%d rows)" % (name, k[0], k[1], v / max(n, 1), n))
PY
