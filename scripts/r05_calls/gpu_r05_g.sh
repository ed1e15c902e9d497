#!/bin/bash
# round 5, seventh device call: stalled histories handed to the sweep (test, and the bench with the bad read planted anywhere), the whole bench line
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 400 python -m pytest tests/test_stall_handover_gpu.py -q -m gpu -p no:cacheprovider --tb=short -x -s 2>&1 | tail -15 > $OUT/handover_test.txt
timeout -k 5 500 python bench.py > $OUT/bench.stdout 2> $OUT/bench.stderr
cp gpurun_out/bench_full.json $OUT/bench_full.json 2>/dev/null
tail -8 $OUT/handover_test.txt; grep -v "^\[bench full\]" $OUT/bench.stderr | tail -4; tail -1 $OUT/bench.stdout
