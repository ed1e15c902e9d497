#!/bin/bash
# round 5, twelfth device call: the shipped-defaults parity file; the headline with first visited sets of 2 and 3 entries per op
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_k
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 600 python -m pytest tests/test_shipped_defaults_gpu.py -q -m gpu -p no:cacheprovider --tb=short 2>&1 | tail -15 > $OUT/defaults_tests.txt
for v in 2 3; do
  timeout -k 5 300 python bench.py --only-headline --visited-per-op $v > $OUT/bench_vpo$v.stdout 2> $OUT/bench_vpo$v.stderr
  tail -1 $OUT/bench_vpo$v.stdout | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('visited per op', '$v', d['value'], d['ms_per_step'], 'search', d['roofline']['kernel_ms'], d['extra'].get('device_ms'), 'GB', d['extra'].get('device_GB_per_batch'), 'probes', d['roofline']['probes_per_launch'])"
done
tail -12 $OUT/defaults_tests.txt
