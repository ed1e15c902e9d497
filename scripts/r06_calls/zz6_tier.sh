# the whole -m gpu tier once more at the round's final HEAD
OUT=gpurun_out/r06_last6
mkdir -p $OUT
timeout 3000 python -m pytest tests -q -m gpu --durations=5 > $OUT/gpu_tests.txt 2>&1
tail -8 $OUT/gpu_tests.txt
