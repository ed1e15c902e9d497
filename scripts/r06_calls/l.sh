set -x
mkdir -p gpurun_out/r06_l
timeout 1500 python -m pytest tests/test_multi_register.py tests/test_baseline_configs.py -x -q -m gpu --durations=8 > gpurun_out/r06_l/mr_tests.txt 2>&1
tail -25 gpurun_out/r06_l/mr_tests.txt
