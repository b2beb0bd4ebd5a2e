# the GPU suite with every test's duration, smoke first
O=gpurun_out/r06_suite; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python -m pytest tests -m gpu -q --durations=0 -x 2>&1 | tail -400 > $O/pytest_gpu.log
grep -v "^[0-9.]*s \(setup\|teardown\)" $O/pytest_gpu.log | head -150
tail -5 $O/pytest_gpu.log
