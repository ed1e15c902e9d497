#!/bin/bash
# The next round's FIRST gpurun call (one box, ~20 minutes): everything round 4 prepared under the emulators and could not measure.
#   1. the GPU tests (the experimental forms are tests/test_zy_forms_gpu.py: XPASS = the form is right on the device)
#   2. bench.py's two form legs alone: one history under every form of the single-history path, a batch under both packs
#   3. rocprofv3 kernel stats of a batch pass under pack_kernel + open_counts_kernel, under pack_wg_kernel (TBC_PACK_WG=2) and over the
#      lean tables (TBC_NARROW_LEAN=1), and with everything at once (lean tables + lazy lookahead + lists in order of completion + fused pack)
# usage (from the repo root on the GPU box):  bash scripts/gpu_round5_first.sh     -> gpurun_out/r05_first/*
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_first
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 900 python -m pytest tests -q -m gpu -rxX 2>&1 | tail -40 > $OUT/gpu_tests.txt
for leg in single_history_forms batch_forms; do
  TBC_BENCH_FORMS_BUDGET_S=900 timeout -k 5 1200 python bench.py --leg $leg 2> $OUT/$leg.stderr | tail -1 > $OUT/$leg.json     # (every form: the driver's run has 80 s a leg)
done
cd /tmp && export TMPDIR=/tmp
for form in default wg lean all all2; do
  case $form in wg) E="TBC_PACK_WG=2";; lean) E="TBC_NARROW_LEAN=1";; all) E="TBC_NARROW_LEAN=2 TBC_NARROW_ORDER=1 TBC_PACK_WG=2";; all2) E="TBC_NARROW_LEAN=2 TBC_NARROW_ORDER=40 TBC_PACK_WG=2";; *) E="TBC_PACK_WG=0";; esac
  env $E timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$form -o p -- \
    python $GRAFT_REPO_ROOT/scripts/gpu_narrow_ab.py 16384 0.1 8 4 3 > $OUT/trace_$form.log 2>&1 < /dev/null
  f=$(ls $OUT/trace_$form/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -12 "$f" > $OUT/kernel_stats_$form.csv
  rm -rf $OUT/trace_$form
done
cd $GRAFT_REPO_ROOT
tail -5 $OUT/gpu_tests.txt; for leg in single_history_forms batch_forms; do python - $OUT/$leg.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    for e in d.get("result", []):
        print({k: e[k] for k in e if k in ("form", "valid_median_ms", "ms_per_pass", "device_ms", "breakdown_us", "counters_match", "error")})
except Exception as ex:
    print("no result:", ex)
PY
done
for form in default wg lean all all2; do echo "== kernel stats, form $form"; cut -c1-160 $OUT/kernel_stats_$form.csv 2>/dev/null; done
