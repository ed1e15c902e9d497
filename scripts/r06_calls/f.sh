#!/bin/bash
# round 6, sixth device call: the race of list orders with first visited sets of 64 entries an op and groups sized to the free memory
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 600 python -m pytest tests/test_order_restarts_gpu.py -q -m gpu -p no:cacheprovider --tb=short 2>&1 | tail -30 > $OUT/gpu_tests.txt
tail -15 $OUT/gpu_tests.txt
TBC_DEBUG=2 timeout -k 5 400 python bench.py --leg workload_3 --no-cpu > $OUT/workload_3.json 2> $OUT/workload_3.stderr; tail -c 900 $OUT/workload_3.json; echo
grep "race\|run: " $OUT/workload_3.stderr | tail -60
TBC_DEBUG=2 timeout -k 5 600 python bench.py --leg workload_2 --no-cpu > $OUT/workload_2.json 2> $OUT/workload_2.stderr; tail -c 900 $OUT/workload_2.json; echo
grep "race" $OUT/workload_2.stderr | tail -40
