#!/bin/bash
# round 6: where pack_wg64_kernel's time goes (per-phase clocks, TBC_PACK_PROF builds) at 8 / 2 workgroups a CU; kernel stats of one batch alone
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_i
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for t in packprof packprof2; do
  echo "== $t"
  TBC_LIB_PATH=$GRAFT_REPO_ROOT/jepsen-tigerbeetle_amd/csrc/variants/libtbcheck_$t.so timeout -k 5 300 python scripts/gpu_pack_prof.py 32768 2>&1 | grep -v "^\[tbc\|amdgpu.ids" | tail -9
done > $OUT/pack_prof_after_fence.txt
cat $OUT/pack_prof_after_fence.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python $GRAFT_REPO_ROOT/scripts/gpu_narrow_ab.py 32768 0.1 8 4 3 > $OUT/trace.log 2>&1 < /dev/null
f=$(ls $OUT/trace/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200
rm -rf $OUT/trace/*kernel_trace.csv
