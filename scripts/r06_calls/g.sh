#!/bin/bash
# round 6, seventh device call: the race of list orders as ONE replica batch beside the RESUMED default order
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 600 python -m pytest tests/test_order_restarts_gpu.py tests/test_limits_gpu.py -q -m gpu -p no:cacheprovider --tb=short 2>&1 | tail -30 > $OUT/gpu_tests.txt
tail -25 $OUT/gpu_tests.txt
TBC_DEBUG=2 timeout -k 5 400 python bench.py --leg workload_3 --no-cpu > $OUT/workload_3.json 2> $OUT/workload_3.stderr; tail -c 900 $OUT/workload_3.json; echo
grep "race" $OUT/workload_3.stderr | tail -12
TBC_DEBUG=2 timeout -k 5 600 python bench.py --leg workload_2 --no-cpu > $OUT/workload_2.json 2> $OUT/workload_2.stderr; tail -c 900 $OUT/workload_2.json; echo
grep "race" $OUT/workload_2.stderr | tail -16
