# the closing evidence of round 6 (one gpurun call): rocprofv3 kernel stats of the bench's timed region, PMC passes with HBM traffic at HEAD
# (-> profiles/r06_traffic.json on the box, copied out through gpurun_out/profiles/), the set-full leg under rocprofv3, the final bench line,
# and the parity files once more under TBC_GUARD=1 (guard bytes behind every device arena).
set -x
OUT=gpurun_out/r06_final
mkdir -p $OUT
bash scripts/gpu_profile_r06.sh 32768 8 r06_final traffic > $OUT/profile_call.txt 2>&1
tail -25 $OUT/profile_call.txt | cut -c1-400
(cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/setfull_trace -o p -- python $GRAFT_REPO_ROOT/bench.py --leg set_full > $GRAFT_REPO_ROOT/$OUT/setfull_leg_under_rocprof.txt 2>&1 < /dev/null)
f=$(ls $OUT/setfull_trace/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -8 "$f" > $OUT/setfull_kernel_stats_head.csv
rm -rf $OUT/setfull_trace
timeout 300 python bench.py --leg set_full > $OUT/setfull_leg.txt 2>&1
tail -1 $OUT/setfull_leg.txt | cut -c1-600
timeout 1200 python bench.py > $OUT/bench_final.json.log 2> $OUT/bench_final.err
tail -c 600 $OUT/bench_final.json.log
cp gpurun_out/bench_full.json $OUT/bench_full_final.json 2>/dev/null
TBC_GUARD=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_stream_gpu.py tests/test_limits_gpu.py tests/test_multi_register.py tests/test_order_restarts_gpu.py tests/test_count_form_gpu.py tests/test_comm_gpu.py -x -q -m gpu -k "not passes_equal_the_oracles_pipeline" > $OUT/gpu_tests_under_guard.txt 2>&1
tail -4 $OUT/gpu_tests_under_guard.txt
