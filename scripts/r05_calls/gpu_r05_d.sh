#!/bin/bash
# round 5, fourth device call: one history after the upload / memset / pinned-readback changes (trace + timing), the single-history GPU tests
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TBC_DEBUG=2 timeout -k 5 120 python scripts/gpu_one_history.py competition 1 > $OUT/one_history_trace.txt 2>&1
timeout -k 5 120 python scripts/gpu_latency.py > $OUT/latency.txt 2>&1
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py tests/test_sweep.py tests/test_count_form_gpu.py tests/test_edn_golden.py tests/test_list_order_gpu.py::test_list_order_where_it_does_not_apply_is_slot_order tests/test_zz_smoke_entry.py -q -m gpu -p no:cacheprovider --tb=short -x 2>&1 | tail -15 > $OUT/gpu_tests.txt
tail -32 $OUT/one_history_trace.txt; tail -12 $OUT/latency.txt; tail -8 $OUT/gpu_tests.txt
