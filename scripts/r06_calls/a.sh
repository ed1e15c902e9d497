#!/bin/bash
# round 6, first device call: the split host library (batch_create / batch_run / batch_shard / batch_stream) under the whole GPU tier,
# the streaming tests, the smoke entry, the default bench line with extra.fresh_input
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 400 python -m pytest tests/test_stream_gpu.py -q -m gpu -p no:cacheprovider --tb=short -x 2>&1 | tail -40 > $OUT/gpu_tests_stream.txt
timeout -k 5 900 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short --deselect tests/test_stream_gpu.py 2>&1 | tail -40 > $OUT/gpu_tests.txt
timeout -k 5 200 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1
timeout -k 5 700 python bench.py > $OUT/bench.stdout 2> $OUT/bench.stderr
cp gpurun_out/bench_full.json $OUT/bench_full.json 2>/dev/null
free -g | head -2 > $OUT/host_mem.txt; nproc >> $OUT/host_mem.txt
echo "== stream tests"; tail -30 $OUT/gpu_tests_stream.txt
echo "== all gpu tests"; tail -15 $OUT/gpu_tests.txt
echo "== smoke"; tail -3 $OUT/smoke.txt
echo "== bench"; grep -v "^\[bench full\]" $OUT/bench.stderr | tail -25; tail -1 $OUT/bench.stdout
