set -x
mkdir -p gpurun_out/r06_m
timeout 900 python -m pytest tests/test_limits_gpu.py -x -q -m gpu --durations=3 > gpurun_out/r06_m/tests.txt 2>&1
tail -8 gpurun_out/r06_m/tests.txt
timeout 900 python bench.py > gpurun_out/r06_m/bench.json.log 2> gpurun_out/r06_m/bench.err
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r06_m/bench.json.log') if x.startswith('{')][-1]
d=json.loads(l)
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['extra']['one_batch_at_a_time'], d['extra']['fresh_input']['histories_per_s'], d['extra']['device_ms'])
PY
