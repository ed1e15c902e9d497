# set-full at HEAD once more (resolve without the butterfly, rows a step ahead): tests, the leg, kernel stats, the counter passes -> profiles/r06_setfull_traffic.json
OUT=gpurun_out/r06_last4
mkdir -p $OUT gpurun_out/profiles
timeout 600 python -m pytest tests/test_set_full.py -q -m gpu > $OUT/tests.txt 2>&1; tail -2 $OUT/tests.txt
bash scripts/gpu_profile_setfull.sh r06_setfull_last > $OUT/setfull_pmc_call.txt 2>&1
python scripts/update_setfull_traffic.py gpurun_out/prof_r06_setfull_last/pmc_summary.txt && cp profiles/r06_setfull_traffic.json gpurun_out/profiles/ && cp gpurun_out/prof_r06_setfull_last/pmc_summary.txt gpurun_out/profiles/r06_setfull_pmc.txt
(cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/setfull_trace -o p -- python $GRAFT_REPO_ROOT/bench.py --leg set_full > $GRAFT_REPO_ROOT/$OUT/setfull_leg_under_rocprof.txt 2>&1 < /dev/null)
f=$(ls $OUT/setfull_trace/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $OUT/setfull_kernel_stats.csv; rm -rf $OUT/setfull_trace
timeout 300 python bench.py --leg set_full 2>/dev/null | tail -1 > $OUT/setfull_leg.json; cut -c1-700 $OUT/setfull_leg.json
