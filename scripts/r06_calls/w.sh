# (the TBC_SETFULL_PAD knob this call scans was an experiment's and is gone from the library: profiles/r06_setfull_pad_scan.txt)
# set-full: a row pitch that is not a multiple of the memory channels' period (pad scan), kernel trace of the best
OUT=gpurun_out/r06_w
mkdir -p $OUT
for pad in 0 16 32 64 128 192 256 320 512 1024 1088; do
  TBC_SETFULL_PAD=$pad timeout 300 python bench.py --leg set_full 2>/dev/null | tail -1 > $OUT/leg_pad$pad.json
  python - <<PY
import json
d=json.load(open("$OUT/leg_pad$pad.json"))["result"]
print("pad", $pad, d["scan_ms"], d["roofline"]["frac"], d["lost_elements_found"])
PY
done
TBC_SETFULL_PAD=64 timeout 600 python -m pytest tests/test_set_full.py -q -m gpu > $OUT/tests_pad64.txt 2>&1; tail -2 $OUT/tests_pad64.txt
for pad in 0 64; do
(cd /tmp && export TMPDIR=/tmp && TBC_SETFULL_PAD=$pad timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace$pad -o p -- python $GRAFT_REPO_ROOT/bench.py --leg set_full > /dev/null 2>&1 < /dev/null)
f=$(ls $OUT/trace$pad/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_pad$pad.csv && grep -i setfull $OUT/kernel_stats_pad$pad.csv
rm -rf $OUT/trace$pad
done
