#!/bin/bash
# round 5: checker/set-full after the summaries went column-major and the streaming pass to eight rows of 16 B a lane: its GPU tests, the
# bench leg, and the kernels' split under rocprofv3
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_r
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 300 python -m pytest tests/test_set_full.py -q -m gpu -p no:cacheprovider --tb=short 2>&1 | tail -15 > $OUT/gpu_tests.txt
timeout -k 5 120 python bench.py --leg set_full > $OUT/leg.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python $GRAFT_REPO_ROOT/bench.py --leg set_full > $OUT/leg_rocprof.txt 2>&1 < /dev/null
f=$(ls $OUT/trace/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && grep -i "setfull\|Name" "$f" > $OUT/kernel_stats_set_full.csv
rm -rf $OUT/trace
tail -5 $OUT/gpu_tests.txt; tail -2 $OUT/leg.txt | cut -c1-900; cat $OUT/kernel_stats_set_full.csv | cut -c1-220
