#!/bin/bash
# round 5, sixteenth device call: one history, longer segments (TBC_SWEEP_SEG 40 / 48 / 56 / 64 / 96), twice
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_o
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 150 python scripts/gpu_latency.py 0.1 40 48 56 64 96 40 48 > $OUT/latency.txt 2>&1
grep "linear" $OUT/latency.txt
