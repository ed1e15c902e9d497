mkdir -p gpurun_out/r06_q
for cfg in "8 32768" "16 32768" "8 8192" "8 16384"; do
set -- $cfg
TBC_BENCH_LEG_LANES=$1 timeout 900 python bench.py --leg workload_crashed --batch4 $2 --no-cpu > gpurun_out/r06_q/crashed_l$1_$2.txt 2> gpurun_out/r06_q/crashed_l$1_$2.err
echo "lanes $1 batch $2"; tail -1 gpurun_out/r06_q/crashed_l$1_$2.txt | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read())['result']; print({k:d[k] for k in ('histories_per_gpu','search_width','lanes_per_history','list_order','value','ms_per_step','valid','unknown')}, d['device_ms'])
except Exception as e: print('failed', e)
"
tail -2 gpurun_out/r06_q/crashed_l$1_$2.err | grep -v amdgpu
done
