#!/bin/bash
# round 6, eighth device call: the whole GPU tier + smoke at this tree
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short 2>&1 | tail -60 > $OUT/gpu_tests.txt
timeout -k 5 200 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1
tail -40 $OUT/gpu_tests.txt; tail -3 $OUT/smoke.txt
