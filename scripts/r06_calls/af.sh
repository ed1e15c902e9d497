# the crashed-batch leg at several batch sizes: 6 wavefronts a SIMD = 6,144 histories resident; is 8,192 one round and a third?
OUT=gpurun_out/r06_ah; mkdir -p $OUT
for b in 6144 8192 12288 18432; do
  timeout 600 python bench.py --leg workload_crashed --batch4 $b --no-cpu 2>/dev/null | tail -1 > $OUT/crashed_$b.json
  python - <<PY
import json
d=json.load(open("$OUT/crashed_$b.json"))["result"]
print($b, d.get("value"), d.get("ms_per_step"), d.get("device_ms"), d.get("unknown"), d.get("error"))
PY
done
