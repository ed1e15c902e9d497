set -x
mkdir -p gpurun_out/r06_l
timeout 900 python -m pytest tests/test_multi_register.py -x -q -m gpu --durations=4 > gpurun_out/r06_l/mr_tests2.txt 2>&1
tail -8 gpurun_out/r06_l/mr_tests2.txt
timeout 900 python scripts/gpu_config4_time.py 0.03 100000 > gpurun_out/r06_l/config4_time.txt 2>&1
tail -4 gpurun_out/r06_l/config4_time.txt
