#!/bin/bash
# round 6, fifth device call: the race of list orders (and the sequential form with a witness) against the oracle; what it buys at real
# concurrency (bench legs workload_3: ~19 calls in flight, workload_2: ~32)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 900 python -m pytest tests/test_order_restarts_gpu.py tests/test_list_order_gpu.py tests/test_shipped_defaults_gpu.py -q -m gpu -p no:cacheprovider --tb=short 2>&1 | tail -60 > $OUT/gpu_tests.txt
tail -40 $OUT/gpu_tests.txt
timeout -k 5 400 python bench.py --leg workload_3 > $OUT/workload_3.json 2> $OUT/workload_3.stderr; tail -c 1600 $OUT/workload_3.json; echo
timeout -k 5 600 python bench.py --leg workload_2 > $OUT/workload_2.json 2> $OUT/workload_2.stderr; tail -c 1600 $OUT/workload_2.json; echo
tail -5 $OUT/workload_2.stderr
