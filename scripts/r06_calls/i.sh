#!/bin/bash
# round 6: where pack_wg64_kernel's time goes (per-phase clocks, TBC_PACK_PROF builds) at 8 / 4 / 2 / 1 workgroups a CU
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_i
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for t in packprof packprof4 packprof2 packprof1; do
  echo "== $t"
  TBC_LIB_PATH=$GRAFT_REPO_ROOT/jepsen-tigerbeetle_amd/csrc/variants/libtbcheck_$t.so timeout -k 5 300 python scripts/gpu_pack_prof.py 32768 2>&1 | grep -v "^\[tbc\|amdgpu.ids" | tail -9
done > $OUT/pack_prof_occupancy.txt
cat $OUT/pack_prof_occupancy.txt
