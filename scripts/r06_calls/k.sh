set -x
mkdir -p gpurun_out/r06_k
for cfg in "32768 2500" "32768 5000" "65536 5000" "16384 10000" "16384 20000" "32768 20000" "8192 10000"; do
set -- $cfg
timeout 900 python scripts/gpu_round_spread.py $1 $2 > gpurun_out/r06_k/spread_$1_$2.txt 2>&1
tail -4 gpurun_out/r06_k/spread_$1_$2.txt | head -2
done
