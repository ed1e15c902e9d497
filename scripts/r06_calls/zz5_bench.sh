# the driver's bench command and the smoke entry once more at the round's final HEAD (set-full's last changes came after zz2_last.sh)
OUT=gpurun_out/r06_last5
mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1200 python bench.py > $OUT/bench.json.log 2> $OUT/bench.err
tail -c 700 $OUT/bench.json.log
cp gpurun_out/bench_full.json $OUT/bench_full.json 2>/dev/null
