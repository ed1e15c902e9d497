#!/bin/bash
# round 5, seventeenth device call: the sweep's tests and the crashed-form tests at the new window length, then the driver's bench command again
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_p
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 300 python -m pytest tests/test_sweep.py tests/test_count_form_gpu.py tests/test_edn_golden.py tests/test_zz_smoke_entry.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider --tb=short -x 2>&1 | tail -5 > $OUT/gpu_tests.txt
timeout -k 5 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.stdout 2> $OUT/bench.stderr
cp gpurun_out/bench_full.json $OUT/bench_full.json 2>/dev/null
tail -3 $OUT/gpu_tests.txt; wc -c $OUT/bench.stdout; tail -1 $OUT/bench.stdout
