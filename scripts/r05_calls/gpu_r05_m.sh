#!/bin/bash
# round 5, fourteenth device call: the shipped-defaults parity file, bounded (every test under pytest-timeout, the call under 240 s)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_m
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 240 python -m pytest tests/test_shipped_defaults_gpu.py -q -m gpu -p no:cacheprovider --tb=short --durations=6 2>&1 | tail -25 > $OUT/defaults_tests.txt
tail -25 $OUT/defaults_tests.txt
