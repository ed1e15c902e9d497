#!/bin/bash
# round 5, tenth device call: the narrow kernel's ring of 8 (5.9 KB of LDS a wavefront): headline two in flight and one batch at a time
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_j
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for f in 2 1 2; do
  timeout -k 5 300 python bench.py --only-headline --in-flight $f > $OUT/bench_f$f.stdout 2> $OUT/bench_f$f.stderr
  tail -1 $OUT/bench_f$f.stdout | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('in flight', d['config']['batches_in_flight'], d['value'], d['ms_per_step'], 'search', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], d['extra'].get('device_ms'))"
done
timeout -k 5 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider --tb=short -x -k "narrow or batches_in_flight" 2>&1 | tail -3
