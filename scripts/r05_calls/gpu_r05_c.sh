#!/bin/bash
# round 5, third device call: the GPU tests the -x run did not reach, kernel stats + PMC traffic of the headline at this tree, one history traced
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 600 python -m pytest tests/test_list_order_gpu.py tests/test_multi_register.py tests/test_set_full.py tests/test_sweep.py tests/test_zz_smoke_entry.py -q -m gpu -p no:cacheprovider --tb=short 2>&1 | tail -30 > $OUT/gpu_tests_rest.txt
TBC_DEBUG=2 timeout -k 5 120 python scripts/gpu_one_history.py competition 1 > $OUT/one_history_trace.txt 2>&1
bash scripts/gpu_profile_r05.sh 32768 8 r05_l8 traffic > $OUT/profile.txt 2>&1
cp gpurun_out/prof_r05_l8/pmc_summary.txt $OUT/pmc_summary.txt 2>/dev/null; cp gpurun_out/prof_r05_l8/kernel_stats_head.csv $OUT/kernel_stats_head.csv 2>/dev/null
tail -12 $OUT/gpu_tests_rest.txt; tail -40 $OUT/one_history_trace.txt; tail -30 $OUT/profile.txt
