# round 6: the ragged per-element units (hq_k_brick_het<., RAGGED>) -- parity tests, then the gradient basins' bench lines
O=gpurun_out/r06_gradient; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_reference_link.py -m gpu -x -q --durations=15 \
  -k "gradient or eight_ranks or cvm_database or quarter or small_lateral or (cone_windows and not c3h) or (eight_partitions and c3)" 2>&1 | tail -40 > $O/pytest.log
cat $O/pytest.log
timeout 300 python bench.py --workload o4gs --no-cpu-baseline --no-pmc > $O/bench_o4gs.json 2>$O/bench_o4gs.err; cut -c1-400 $O/bench_o4gs.json; tail -3 $O/bench_o4gs.err
timeout 900 python bench.py --workload o4g --no-cpu-baseline --pmc-dir $O/pmc_o4g > $O/bench_o4g.json 2>$O/bench_o4g.err; cut -c1-400 $O/bench_o4g.json; tail -3 $O/bench_o4g.err
