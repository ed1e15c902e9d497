#!/bin/bash
# round 5, second device call: the whole GPU tier + smoke under the flipped defaults, then the headline bench
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 900 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short -x 2>&1 | tail -40 > $OUT/gpu_tests.txt
timeout -k 5 200 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1
timeout -k 5 500 python bench.py > $OUT/bench.stdout 2> $OUT/bench.stderr
cp gpurun_out/bench_full.json $OUT/bench_full.json 2>/dev/null
tail -15 $OUT/gpu_tests.txt; tail -3 $OUT/smoke.txt; grep -v "^\[bench full\]" $OUT/bench.stderr | tail -5; tail -1 $OUT/bench.stdout
