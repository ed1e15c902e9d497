#!/bin/bash
# round 5, sixth device call: the relaxed sweep BESIDE the exact search (abort word): tests three times over, the crashed tiers
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  timeout -k 5 400 python -m pytest tests/test_count_form_gpu.py -q -m gpu -p no:cacheprovider --tb=short -x -k "relaxed_sweep or tiers" 2>&1 | tail -12 > $OUT/gpu_tests_$i.txt
done
timeout -k 5 300 python -m pytest tests/test_count_form_gpu.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider --tb=short -x 2>&1 | tail -8 > $OUT/gpu_tests_all.txt
timeout -k 5 300 python bench.py --leg tiers > $OUT/tiers.json 2> $OUT/tiers.stderr
for i in 1 2 3; do tail -3 $OUT/gpu_tests_$i.txt; done; tail -4 $OUT/gpu_tests_all.txt
python - <<'PY'
import json, os
d = json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r05_f/tiers.json")).read().strip().splitlines()[-1])
for t in d["result"]:
    print(t["info_rate"], t["history"], "gpu_ms", t["gpu_ms"], "verdict", t["gpu_verdict"], "cpu_port", t["cpu_port_ms"], "cpu_same", t["cpu_same_algorithm_ms"], t["cpu_same_algorithm_passes"])
PY
