#!/bin/bash
# One history through tbc_check under every form of the level sweep (run on the GPU box via gpurun; ~10 s each): K6, K6w as measured
# in round 4, and the forms prepared under the emulator but not yet measured (ring, fingerprint, both).  The counters printed must be the
# same line in every run.  Then the sweep's GPU tests under the fastest of them.   usage: scripts/gpu_sweep_wg_variants.sh [out.log]
OUT=${1:-gpurun_out/sweep_wg_variants.log}
: > $OUT
run() { echo "== $*" >> $OUT; env "$@" timeout 60 python scripts/gpu_sweep_wg.py 2>&1 | grep -v amdgpu.ids >> $OUT; }
run TBC_SWEEP_WG=0
run TBC_SWEEP_WG=8
run TBC_SWEEP_WG=8 TBC_SWEEP_WG_RING=1
run TBC_SWEEP_WG=8 TBC_SWEEP_WG_FP=1
run TBC_SWEEP_WG=8 TBC_SWEEP_WG_RING=1 TBC_SWEEP_WG_FP=1
run TBC_SWEEP_WG=8 TBC_SWEEP_WG_FP=1 TBC_SWEEP_SEG=16
run TBC_SWEEP_WG=16
cat $OUT
for v in "TBC_SWEEP_WG_RING=1" "TBC_SWEEP_WG_FP=1" "TBC_SWEEP_WG_RING=1 TBC_SWEEP_WG_FP=1"; do
  echo "== tests/test_sweep.py under $v" | tee -a $OUT
  env $v timeout 200 python -m pytest tests/test_sweep.py -x -q -m gpu 2>&1 | tail -2 | tee -a $OUT
done
