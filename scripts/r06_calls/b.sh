#!/bin/bash
# round 6, second device call: the tests written since call a -- time limits firing in every kernel, :configs of count-form invalid
# verdicts, the exchange behind the C-ABI (two processes over a host transport; RCCL at world 1), set-full lists in any order
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 900 python -m pytest tests/test_limits_gpu.py tests/test_comm_gpu.py tests/test_set_full.py tests/test_count_form_gpu.py tests/test_stream_gpu.py -q -m gpu -p no:cacheprovider --tb=short 2>&1 | tail -80 > $OUT/gpu_tests.txt
tail -70 $OUT/gpu_tests.txt
