# (the TBC_SETFULL_CHUNKS knob this call scans was an experiment's and is gone from the library: profiles/NOTES_r06.md)
# set-full: the streaming pass with the chunk in blockIdx.x (XCD balance); GPU tests, the bench leg at several chunk counts, a kernel trace
OUT=gpurun_out/r06_v
mkdir -p $OUT
timeout 600 python -m pytest tests/test_set_full.py -q -m gpu > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt
for c in default 128 256 512 1024; do
  for rep in 1 2; do
    if [ $c = default ]; then timeout 300 python bench.py --leg set_full 2>/dev/null | tail -1 > $OUT/leg_$c.$rep.json
    else TBC_SETFULL_CHUNKS=$c timeout 300 python bench.py --leg set_full 2>/dev/null | tail -1 > $OUT/leg_$c.$rep.json; fi
    python - <<PY
import json
d=json.load(open("$OUT/leg_$c.$rep.json"))["result"]
print("$c", "$rep", d["scan_ms"], d["roofline"]["frac"], d["bytes_scanned"])
PY
  done
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o sf -- python $GRAFT_REPO_ROOT/bench.py --leg set_full > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/setfull_kernel_stats.csv
grep -i "setfull" $OUT/setfull_kernel_stats.csv | head
rm -rf $OUT/prof
