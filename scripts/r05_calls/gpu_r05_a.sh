#!/bin/bash
# round 5, first device call: why tests/test_lean_gpu.py fails under TBC_NARROW_ORDER; the single-history forms' timings; the default bench line
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for ord in 1 2 40; do
  TBC_NARROW_ORDER=$ord timeout -k 5 240 python -m pytest tests/test_lean_gpu.py -q -m gpu -p no:cacheprovider --tb=short 2>&1 | tail -60 > $OUT/lean_gpu_order_$ord.txt
done
TBC_BENCH_FORMS_BUDGET_S=150 timeout -k 5 300 python bench.py --leg single_history_forms 2> $OUT/single_history_forms.stderr | tail -1 > $OUT/single_history_forms.json
timeout -k 5 400 python bench.py > $OUT/bench_default.stdout 2> $OUT/bench_default.stderr
tail -c 600 $OUT/bench_default.stderr | head -c 0
cp gpurun_out/bench_full.json $OUT/bench_full.json 2>/dev/null
for ord in 1 2 40; do echo "== order $ord"; tail -25 $OUT/lean_gpu_order_$ord.txt; done
python - $OUT/single_history_forms.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    for e in d.get("result", []):
        print({k: e[k] for k in e if k in ("form", "valid_median_ms", "valid_min_ms", "breakdown_us", "counters_match", "error", "skipped")})
except Exception as ex:
    print("no result:", ex)
PY
wc -c $OUT/bench_default.stdout; tail -1 $OUT/bench_default.stdout
