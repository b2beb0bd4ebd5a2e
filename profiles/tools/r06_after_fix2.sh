# the basins' lines with timed batches that enqueue exactly what hq_run does (no hold-back of the compute stream)
O=gpurun_out/r06_fix2; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_bench.py -m gpu -x -q -k "small_lateral or gradient or lines_of or one_rank" 2>&1 | tail -5
for w in o4 o3 o4g c3; do python bench.py --workload $w --no-cpu-baseline --no-pmc > $O/bench_$w.json 2>/dev/null; echo $w; python3 -c "
import json; d=json.load(open('$O/bench_$w.json')); c=d['config']; r=d['roofline']; print(round(d['ms_per_step'],4), round(d['value']/1e9,2), c['ms_per_step_runs'], c['parity_windows'], c['parity_worst'], 'frac', round(r['frac'],3), 'kernel_ms', round(r['kernel_ms'],4), r['phase_us'])"; done
