# the parity files once more under TBC_GUARD=1 (guard bytes behind every device arena) at HEAD, set-full's among them
OUT=gpurun_out/r06_last3
mkdir -p $OUT
TBC_GUARD=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_stream_gpu.py tests/test_limits_gpu.py tests/test_multi_register.py tests/test_order_restarts_gpu.py tests/test_count_form_gpu.py tests/test_comm_gpu.py tests/test_set_full.py -x -q -m gpu -k "not passes_equal_the_oracles_pipeline" > $OUT/gpu_tests_under_guard.txt 2>&1
tail -4 $OUT/gpu_tests_under_guard.txt
