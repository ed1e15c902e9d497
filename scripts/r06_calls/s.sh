mkdir -p gpurun_out/r06_s
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --only-headline 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['extra'].get('device_ms'))" | tee -a gpurun_out/r06_s/search_priority_ab.txt; }
run "search low priority F2" TBC_SEARCH_PRIORITY=-1
run "search high priority F2" TBC_SEARCH_PRIORITY=1
run "default F2" TBC_SEARCH_PRIORITY=0
run "search low priority F3" TBC_SEARCH_PRIORITY=-1 TBC_BENCH_IN_FLIGHT=3
