#!/bin/bash
# round 6, fourth device call: ORDER RESTARTS -- parity with the oracle's pipeline, the tests that pin the library's defaults, and what they
# buy at real concurrency (bench legs workload_2: ~32 calls in flight, workload_3: ~19); 4 against 8 lanes per history on the headline
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 900 python -m pytest tests/test_order_restarts_gpu.py tests/test_list_order_gpu.py tests/test_shipped_defaults_gpu.py tests/test_zz_smoke_entry.py -q -m gpu -p no:cacheprovider --tb=short 2>&1 | tail -60 > $OUT/gpu_tests.txt
tail -40 $OUT/gpu_tests.txt
timeout -k 5 400 python bench.py --leg workload_3 > $OUT/workload_3.json 2> $OUT/workload_3.stderr; tail -c 1500 $OUT/workload_3.json; echo
timeout -k 5 600 python bench.py --leg workload_2 > $OUT/workload_2.json 2> $OUT/workload_2.stderr; tail -c 1500 $OUT/workload_2.json; echo
timeout -k 5 300 python scripts/gpu_narrow_ab.py 32768 0.1 4,8 4 3 > $OUT/ab_lanes4.txt 2>&1; grep "lanes\|run" $OUT/ab_lanes4.txt | cut -c1-260
