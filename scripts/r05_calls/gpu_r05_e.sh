#!/bin/bash
# round 5, fifth device call: the relaxed sweep in front of the count-form search (tests, the crashed tiers), one history after the cached pinned region
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 700 python -m pytest tests/test_count_form_gpu.py -q -m gpu -p no:cacheprovider --tb=short -x 2>&1 | tail -25 > $OUT/gpu_tests.txt
timeout -k 5 300 python bench.py --leg tiers > $OUT/tiers.json 2> $OUT/tiers.stderr
TBC_DEBUG=2 timeout -k 5 120 python scripts/gpu_one_history.py competition 1 > $OUT/one_history_trace.txt 2>&1
timeout -k 5 120 python scripts/gpu_latency.py > $OUT/latency.txt 2>&1
tail -12 $OUT/gpu_tests.txt; tail -3 $OUT/tiers.stderr; tail -1 $OUT/tiers.json | cut -c1-3000; tail -24 $OUT/one_history_trace.txt; tail -5 $OUT/latency.txt
