#!/bin/bash
# round 5, fifteenth device call: one history -- the spin wait, and the segment length (TBC_SWEEP_SEG 24 / 28 / 32 / 40)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_n
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 150 python scripts/gpu_latency.py 0.1 auto 24 28 40 > $OUT/latency.txt 2>&1
TBC_DEBUG=2 timeout -k 5 60 python scripts/gpu_one_history.py competition 1 > $OUT/trace.txt 2>&1
grep "linear" $OUT/latency.txt; tail -22 $OUT/trace.txt
