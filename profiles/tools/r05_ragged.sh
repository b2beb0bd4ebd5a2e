# ragged brick units (HQ_BK_RAGGED): parity on the small lateral basin, then A / B on one box
O=gpurun_out/r05_ragged; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "lateral_basin" --durations=8 2>&1 | tail -25 > $O/pytest_lateral.log
cat $O/pytest_lateral.log
for wl in o4 o4s; do
  for v in 0 1; do
    HQ_BRICK_RAGGED=$v python bench.py --workload $wl --no-cpu-baseline --no-pmc > $O/bench_${wl}_ragged$v.json 2> $O/bench_${wl}_ragged$v.err
    cut -c1-420 $O/bench_${wl}_ragged$v.json; tail -2 $O/bench_${wl}_ragged$v.err
  done
done
for mf in 64 192 256; do
  HQ_BRICK_RAGGED_MINFILL=$mf python bench.py --workload o4 --no-cpu-baseline --no-pmc --no-parity > $O/bench_o4_minfill$mf.json 2>/dev/null
  cut -c1-300 $O/bench_o4_minfill$mf.json
done
python bench.py --workload o4 --no-cpu-baseline --pmc-dir $O/pmc_o4 > $O/bench_o4_pmc.json 2>/dev/null; cut -c1-200 $O/bench_o4_pmc.json
python bench.py --no-cpu-baseline --no-pmc > $O/bench_c3.json 2>/dev/null; cut -c1-300 $O/bench_c3.json
python bench.py --workload o3 --no-cpu-baseline --no-pmc > $O/bench_o3.json 2>/dev/null; cut -c1-300 $O/bench_o3.json
