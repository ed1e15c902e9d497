set -x
mkdir -p gpurun_out/r06_n
for cfg in "1 2" "0 2" "0 3" "1 3"; do
set -- $cfg
TBC_SEARCH_TURN=$1 timeout 600 python bench.py --in-flight $2 --only-headline --no-cpu --fresh-batches 0 > gpurun_out/r06_n/bench_turn$1_f$2.json.log 2> gpurun_out/r06_n/err_$1_$2.txt || tail -3 gpurun_out/r06_n/err_$1_$2.txt
python - $1 $2 <<'PY'
import json, sys
try:
    l=[x for x in open(f'gpurun_out/r06_n/bench_turn{sys.argv[1]}_f{sys.argv[2]}.json.log') if x.startswith('{')][-1]
    d=json.loads(l)
    print("turn", sys.argv[1], "in flight", sys.argv[2], {k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['extra'].get('device_ms'))
except Exception as e:
    print("failed", sys.argv[1:], e)
PY
done
