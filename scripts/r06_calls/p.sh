mkdir -p gpurun_out/r06_p
timeout 3000 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/r06_p/gpu_tests.txt 2>&1
tail -22 gpurun_out/r06_p/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
