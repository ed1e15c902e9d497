#!/bin/bash
# round 5, ninth device call: the guard-byte hunt again (per-batch check, synchronous poison), the headline at 65,536 histories in ONE batch
# (a finished group takes the next history: the tail of the slowest history amortised), the default headline
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_i
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TBC_GUARD=1 timeout -k 5 300 python bench.py --only-headline --steps 24 > $OUT/guard_bench.stdout 2> $OUT/guard_bench.stderr
grep -c "tbc guard" $OUT/guard_bench.stderr > $OUT/guard_count.txt
TBC_GUARD=1 timeout -k 5 300 python -m pytest tests/test_stall_handover_gpu.py tests/test_zz_smoke_entry.py -q -m gpu -p no:cacheprovider --tb=short -x 2>&1 | tail -4 > $OUT/guard_tests.txt
timeout -k 5 300 python bench.py --only-headline --batch 65536 --in-flight 1 --steps 8 --warmup 1 > $OUT/bench_65536.stdout 2> $OUT/bench_65536.stderr
timeout -k 5 300 python bench.py --only-headline --batch 49152 --in-flight 1 --steps 8 --warmup 1 > $OUT/bench_49152.stdout 2> $OUT/bench_49152.stderr
timeout -k 5 300 python bench.py --only-headline > $OUT/bench_default.stdout 2> $OUT/bench_default.stderr
echo "guard lines:"; cat $OUT/guard_count.txt; grep "tbc guard" $OUT/guard_bench.stderr | head -5; tail -1 $OUT/guard_bench.stdout | cut -c1-300; tail -3 $OUT/guard_tests.txt
for f in bench_65536 bench_49152 bench_default; do echo "== $f"; tail -1 $OUT/$f.stdout | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['extra'].get('device_ms'), d['extra'].get('device_GB_per_batch'))"; done
