#!/bin/bash
# round 5, last device call: the library as rebuilt at HEAD (comment-only changes since the closing call) through the smoke entry and the
# set-full / list-order GPU tests
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_last
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 60 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1
timeout -k 5 80 python -m pytest tests/test_set_full.py tests/test_list_order_gpu.py -q -m gpu -p no:cacheprovider --tb=short 2>&1 | tail -5 > $OUT/gpu_tests.txt
tail -1 $OUT/smoke.txt; tail -2 $OUT/gpu_tests.txt
