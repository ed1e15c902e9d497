mkdir -p gpurun_out/r06_r
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --only-headline 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['extra'].get('device_ms'))" | tee -a gpurun_out/r06_r/residency_ab.txt; }
V=$PWD/jepsen-tigerbeetle_amd/csrc/variants
run "mw3 F2" TBC_LIB_PATH=$V/libtbcheck_mw3.so TBC_BENCH_IN_FLIGHT=2
run "mw3 F3" TBC_LIB_PATH=$V/libtbcheck_mw3.so TBC_BENCH_IN_FLIGHT=3
run "default launched at 3 waves F2" TBC_NARROW_WAVES_PER_SIMD=3 TBC_BENCH_IN_FLIGHT=2
run "default launched at 3 waves F3" TBC_NARROW_WAVES_PER_SIMD=3 TBC_BENCH_IN_FLIGHT=3
run "default F2" TBC_BENCH_IN_FLIGHT=2
