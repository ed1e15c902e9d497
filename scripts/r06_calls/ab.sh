# set-full objects in one allocation (create + run + destroy end to end); the workload_3 leg with two batches in flight
OUT=gpurun_out/r06_ab
mkdir -p $OUT
timeout 600 python -m pytest tests/test_set_full.py -q -m gpu > $OUT/tests.txt 2>&1; tail -2 $OUT/tests.txt
for rep in 1 2; do
  timeout 300 python bench.py --leg set_full 2>/dev/null | tail -1 > $OUT/leg.$rep.json
  python - <<PY
import json
d=json.load(open("$OUT/leg.$rep.json"))["result"]
print("leg", $rep, d["scan_ms"], d["roofline"]["frac"], "end to end", d["end_to_end_ms"])
PY
done
timeout 900 python bench.py --leg workload_3 2>$OUT/w3.err | tail -1 > $OUT/w3.json
python - <<PY
import json
d=json.load(open("$OUT/w3.json"))["result"]
print("workload_3", d.get("value"), d.get("two_in_flight"), d.get("unknown"), d.get("error"))
PY
