# (build variants sfw6 = scripts/build_variant.sh with SRC=set_full -DSF_RESOLVE_MIN_WAVES=6, since made the default)
# set-full resolve with the transposed reductions, at 8 (default build), 6 and <= 6 wavefronts a SIMD
OUT=gpurun_out/r06_af; mkdir -p $OUT
timeout 600 python -m pytest tests/test_set_full.py -q -m gpu > $OUT/tests.txt 2>&1; tail -2 $OUT/tests.txt
for v in default sfw6; do
LIB=$GRAFT_REPO_ROOT/jepsen-tigerbeetle_amd/csrc/variants/libtbcheck_$v.so; [ $v = default ] && LIB=$GRAFT_REPO_ROOT/jepsen-tigerbeetle_amd/csrc/libtbcheck.so
(cd /tmp && export TMPDIR=/tmp && TBC_LIB_PATH=$LIB timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace$v -o p -- python $GRAFT_REPO_ROOT/bench.py --leg set_full > $GRAFT_REPO_ROOT/$OUT/leg$v.txt 2>&1 < /dev/null)
f=$(ls $OUT/trace$v/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_$v.csv; rm -rf $OUT/trace$v
echo "variant $v"; grep resolve $OUT/kernel_stats_$v.csv | sed 's/.*)",//'; tail -1 $OUT/leg$v.txt | python -c "import sys,json; d=json.loads(sys.stdin.read())['result']; print(d['scan_ms'], d['roofline']['frac'], d['lost_elements_found'])"
done
