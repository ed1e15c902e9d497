#!/bin/bash
# round 5, eighth device call: the handover test and the whole bench at threshold 64; then the guard-byte hunt (TBC_GUARD=1): the narrow / wide /
# sweep GPU tests and the headline's passes with 256 poisoned bytes behind every arena
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 400 python -m pytest tests/test_stall_handover_gpu.py -q -m gpu -p no:cacheprovider --tb=short -x -s 2>&1 | tail -6 > $OUT/handover_test.txt
timeout -k 5 500 python bench.py > $OUT/bench.stdout 2> $OUT/bench.stderr
cp gpurun_out/bench_full.json $OUT/bench_full.json 2>/dev/null
TBC_GUARD=1 timeout -k 5 500 python -m pytest tests/test_gpu_parity.py tests/test_list_order_gpu.py tests/test_sweep.py tests/test_count_form_gpu.py -q -m gpu -p no:cacheprovider --tb=short -x 2>&1 | tail -8 > $OUT/guard_tests.txt
TBC_GUARD=1 timeout -k 5 300 python bench.py --only-headline --steps 24 > $OUT/guard_bench.stdout 2> $OUT/guard_bench.stderr
grep -c "tbc guard" $OUT/guard_bench.stderr > $OUT/guard_count.txt
tail -4 $OUT/handover_test.txt; grep -v "^\[bench full\]" $OUT/bench.stderr | tail -3; tail -1 $OUT/bench.stdout; tail -4 $OUT/guard_tests.txt; echo "guard lines in the bench run:"; cat $OUT/guard_count.txt; grep "tbc guard" $OUT/guard_bench.stderr | head -5; tail -1 $OUT/guard_bench.stdout | cut -c1-400
