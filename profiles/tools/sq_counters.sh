#!/bin/bash
# SQ counters of the step kernels of one workload (two rocprofv3 --pmc passes, counters only: no tracing flags):
#   sq_counters.sh <workload> <tag>     -> gpurun_out/r03/sq_<tag>.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
wl=${1:-c2h}; tag=${2:-$wl}
mkdir -p gpurun_out/r03
rm -rf /tmp/sq_$tag; mkdir -p /tmp/sq_$tag
p=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"; do
  p=$((p+1))
  rocprofv3 --pmc $set -d /tmp/sq_$tag/p$p -o out --output-format csv -- python3 bench.py --workload $wl --steps 6 --warmup 2 --no-pmc --no-cpu-baseline > /tmp/sq_$tag/p$p.log 2>&1
done
python3 - "$tag" <<'PY' > gpurun_out/r03/sq_$1_$2.txt 2>&1
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob('/tmp/sq_%s/p*/**/*counter_collection.csv' % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if not k.startswith('hq_k_'): continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] in ('SQ_WAVES', 'SQ_INSTS_VALU'): n[(k, r['Counter_Name'])] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0)):
    print(k, 'dispatches', n[(k, 'SQ_WAVES')])
    wc = c.get('SQ_WAVE_CYCLES', 0) or 1
    for name in sorted(c):
        print('   %-24s %16.0f   %6.3f of WAVE_CYCLES' % (name, c[name], c[name] / wc))
PY
cat gpurun_out/r03/sq_$1_$2.txt
