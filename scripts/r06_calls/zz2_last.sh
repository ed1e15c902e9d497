# the round's last call, second edition (after set-full's rework): set-full's counter passes at HEAD (-> profiles/r06_setfull_traffic.json,
# copied out through gpurun_out/profiles/), its leg under rocprofv3, the whole -m gpu tier, the smoke entry, the driver's bench command
OUT=gpurun_out/r06_last2
mkdir -p $OUT gpurun_out/profiles
bash scripts/gpu_profile_setfull.sh r06_setfull_last > $OUT/setfull_pmc_call.txt 2>&1
python scripts/update_setfull_traffic.py gpurun_out/prof_r06_setfull_last/pmc_summary.txt && cp profiles/r06_setfull_traffic.json gpurun_out/profiles/ && cp gpurun_out/prof_r06_setfull_last/pmc_summary.txt gpurun_out/profiles/r06_setfull_pmc.txt
(cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/setfull_trace -o p -- python $GRAFT_REPO_ROOT/bench.py --leg set_full > $GRAFT_REPO_ROOT/$OUT/setfull_leg_under_rocprof.txt 2>&1 < /dev/null)
f=$(ls $OUT/setfull_trace/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $OUT/setfull_kernel_stats.csv; rm -rf $OUT/setfull_trace
timeout 3000 python -m pytest tests -q -m gpu --durations=8 > $OUT/gpu_tests.txt 2>&1
tail -14 $OUT/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1200 python bench.py > $OUT/bench.json.log 2> $OUT/bench.err
tail -c 900 $OUT/bench.json.log
cp gpurun_out/bench_full.json $OUT/bench_full.json 2>/dev/null
