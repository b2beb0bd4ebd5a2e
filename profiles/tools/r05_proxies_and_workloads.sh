O=gpurun_out/r05_b8; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
HQ_TRACE_TIME_STEPS=1000 python profiles/tools/rank_alone_trace.py 3 40 > $O/rank_alone_wallclock.txt 2>&1
HQ_BRICK_NO_FACES=1 HQ_TRACE_TIME_STEPS=1000 python profiles/tools/rank_alone_trace.py 3 40 > $O/rank_alone_wallclock_nofaces.txt 2>&1
python bench.py --inproc-parts 8 --workload c3 --steps 100 --warmup 10 > $O/inproc8_c3.json 2>$O/inproc8_c3.err
python bench.py --inproc-parts 8 --workload o3 --steps 50 --warmup 5 > $O/inproc8_o3.json 2>$O/inproc8_o3.err
( cd /tmp && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -w -o /tmp/march_twostep $GRAFT_REPO_ROOT/profiles/micro/march_twostep.hip && timeout 600 /tmp/march_twostep > $GRAFT_REPO_ROOT/$O/march_twostep.txt 2>&1 )
( cd /tmp && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -w -o /tmp/march_stencil $GRAFT_REPO_ROOT/profiles/micro/march_stencil.hip && timeout 600 /tmp/march_stencil 2>&1 | head -6 > $GRAFT_REPO_ROOT/$O/march_onestep.txt )
python bench.py --no-cpu-baseline --no-pmc > $O/bench_c3.json 2>/dev/null
python bench.py --workload c2 --no-cpu-baseline --no-pmc > $O/bench_c2.json 2>/dev/null
python bench.py --workload o3 --no-cpu-baseline --no-pmc > $O/bench_o3.json 2>/dev/null
python bench.py --workload o4 --no-cpu-baseline > $O/bench_o4.json 2>/dev/null
python bench.py --workload m1 --no-cpu-baseline --no-pmc > $O/bench_m1.json 2>/dev/null
python bench.py --workload o3s --no-cpu-baseline --no-pmc > $O/bench_o3s.json 2>/dev/null
python bench.py --workload o4s --no-cpu-baseline --no-pmc > $O/bench_o4s.json 2>/dev/null
python bench.py --workload c2h --no-cpu-baseline --no-pmc > $O/bench_c2h.json 2>/dev/null
tail -4 $O/rank_alone_wallclock.txt $O/rank_alone_wallclock_nofaces.txt; cat $O/inproc8_c3.json $O/inproc8_o3.json; cat $O/march_twostep.txt $O/march_onestep.txt
for f in $O/bench_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d['value']/1e9, d['config']['brick_nodes'], d['config']['patches'], d['config'].get('parity_worst'), d['roofline']['kernel_ms'], d['roofline']['frac'])"; done
