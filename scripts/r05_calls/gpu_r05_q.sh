#!/bin/bash
# round 5, after the pack changes (walk places from register keys, pack_wg64, set-full streaming pass): the whole GPU tier, the smoke
# entry, kernel stats + PMC passes of the headline (profiles/r05_traffic.json re-keyed to this tree), then the driver's bench command
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_q
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 600 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short 2>&1 | tail -25 > $OUT/gpu_tests.txt
timeout -k 5 200 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1
bash scripts/gpu_profile_r05.sh 32768 8 r05_q traffic > $OUT/profile.txt 2>&1
cp gpurun_out/prof_r05_q/pmc_summary.txt $OUT/pmc_summary.txt 2>/dev/null; cp gpurun_out/prof_r05_q/kernel_stats_head.csv $OUT/kernel_stats_head.csv 2>/dev/null
cd $GRAFT_REPO_ROOT
timeout -k 5 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.stdout 2> $OUT/bench.stderr
cp gpurun_out/bench_full.json $OUT/bench_full.json 2>/dev/null
tail -4 $OUT/gpu_tests.txt; tail -2 $OUT/smoke.txt; wc -c $OUT/bench.stdout; tail -1 $OUT/bench.stdout; tail -3 $OUT/profile.txt | cut -c1-300; cat $OUT/kernel_stats_head.csv | cut -c1-200
