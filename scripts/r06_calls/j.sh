#!/bin/bash
# round 6: workgroup-scope fences instead of agent-scope ones (no buffer_wbl2 / buffer_inv in pack_wg64, wgl_narrow, wgl_beam):
# the headline batch alone (pack and search ms), parity of the kernels touched, the default bench line
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_j
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 300 python scripts/gpu_narrow_ab.py 32768 0.1 8 4 4 > $OUT/ab_wg_fence.txt 2>&1; grep "lanes\|run" $OUT/ab_wg_fence.txt | cut -c1-300
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_stream_gpu.py tests/test_count_form_gpu.py tests/test_order_restarts_gpu.py tests/test_baseline_configs.py -q -m gpu -p no:cacheprovider --tb=short -x 2>&1 | tail -12 > $OUT/gpu_tests.txt
tail -8 $OUT/gpu_tests.txt
timeout -k 5 900 python bench.py > $OUT/bench.stdout 2> $OUT/bench.stderr
cp gpurun_out/bench_full.json $OUT/bench_full.json 2>/dev/null
grep -v "^\[bench full\]" $OUT/bench.stderr | tail -4; tail -1 $OUT/bench.stdout
