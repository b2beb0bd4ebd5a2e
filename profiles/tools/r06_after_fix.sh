# round 6: behind the dependency fix (the next step's bricks wait for the hanging-node assignment): the octree tests,
# the three basins' bench lines again (their timed regions run through hq_run_timed), the restored full-size partition test
O=gpurun_out/r06_fix; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q --durations=12 -k "lateral or gradient or basin or c2" 2>&1 | tail -25 > $O/pytest.log; cat $O/pytest.log
for w in o4 o3 o4g; do python bench.py --workload $w --no-cpu-baseline --no-pmc > $O/bench_$w.json 2>/dev/null; echo $w; python3 -c "
import json; d=json.load(open('$O/bench_$w.json')); c=d['config']; print(round(d['ms_per_step'],4), round(d['value']/1e9,2), c['ms_per_step_runs'], c['parity_windows'], c['parity_worst'], d['roofline']['phase_us'])"; done
