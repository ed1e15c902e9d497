#!/bin/bash
# round 5, the closing device call at HEAD (after the pack and set-full changes; kernel_sha as in profiles/r05_traffic.json): the whole
# GPU tier, the smoke entry, the driver's bench command, then 50 passes of the headline under TBC_GUARD=1 (the new pack kernels' arenas)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_final2
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 420 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short 2>&1 | tail -25 > $OUT/gpu_tests.txt
timeout -k 5 100 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1
timeout -k 5 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.stdout 2> $OUT/bench.stderr
cp gpurun_out/bench_full.json $OUT/bench_full.json 2>/dev/null
TBC_GUARD=1 timeout -k 5 100 python bench.py --only-headline --steps 48 --warmup 2 > $OUT/guard.stdout 2> $OUT/guard.stderr
grep -ci "guard" $OUT/guard.stderr > $OUT/guard_lines.txt
tail -4 $OUT/gpu_tests.txt; tail -2 $OUT/smoke.txt; wc -c $OUT/bench.stdout; tail -1 $OUT/bench.stdout; echo "guard lines: $(cat $OUT/guard_lines.txt)"; tail -1 $OUT/guard.stdout | cut -c1-400
