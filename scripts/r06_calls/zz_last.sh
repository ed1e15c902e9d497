# the round's last call: the whole -m gpu tier, the smoke entry, and the driver's bench command at HEAD
OUT=gpurun_out/r06_last
mkdir -p $OUT
timeout 3000 python -m pytest tests -q -m gpu --durations=8 > $OUT/gpu_tests.txt 2>&1
tail -14 $OUT/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1200 python bench.py > $OUT/bench.json.log 2> $OUT/bench.err
tail -c 700 $OUT/bench.json.log
cp gpurun_out/bench_full.json $OUT/bench_full.json 2>/dev/null
