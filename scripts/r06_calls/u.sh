mkdir -p gpurun_out/r06_u
for bud in 64 96 128; do
for w in workload_3 workload_2; do
TBC_RESTART_BUDGET=$bud timeout 900 python bench.py --leg $w --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['result']; print('budget $bud $w', d['value'], d['raced_in_six_orders'], d['device_ms'])" | tee -a gpurun_out/r06_u/restart_budget.txt
done; done
