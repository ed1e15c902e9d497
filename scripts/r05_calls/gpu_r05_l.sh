#!/bin/bash
# round 5, thirteenth device call: the shipped-defaults parity file again
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_l
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 600 python -m pytest tests/test_shipped_defaults_gpu.py -q -m gpu -p no:cacheprovider --tb=short --durations=5 2>&1 | tail -25 > $OUT/defaults_tests.txt
tail -25 $OUT/defaults_tests.txt
