# set-full: the lean path of the streaming pass (tiles whose columns all count in every row), GPU tests, the leg, a kernel trace
OUT=gpurun_out/r06_y8
mkdir -p $OUT
timeout 600 python -m pytest tests/test_set_full.py -q -m gpu > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt
for rep in 1 2 3; do
  timeout 300 python bench.py --leg set_full 2>/dev/null | tail -1 > $OUT/leg.$rep.json
  python - <<PY
import json
d=json.load(open("$OUT/leg.$rep.json"))["result"]
print("leg", $rep, d["scan_ms"], d["roofline"]["frac"], d["bytes_scanned"], d["lost_elements_found"])
PY
done
(cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -o p -- python $GRAFT_REPO_ROOT/bench.py --leg set_full > /dev/null 2>&1 < /dev/null)
f=$(ls $OUT/trace/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv && grep -i setfull $OUT/kernel_stats.csv | cut -c1-60,300-
rm -rf $OUT/trace
