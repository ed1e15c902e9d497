#!/bin/bash
# round 6, third device call: forms of the several-histories-per-wavefront search on the bench workload, one batch of 32,768 alone:
# non-temporal visited-set accesses (nt1), the bucket loaded with the first trip (eb1) or touched with it (eb2), both
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
V=jepsen-tigerbeetle_amd/csrc/variants
for t in base nt1 eb1 eb2 eb1nt eb2nt; do
  if [ $t = base ]; then unset TBC_LIB_PATH; else export TBC_LIB_PATH=$GRAFT_REPO_ROOT/$V/libtbcheck_$t.so; fi
  timeout -k 5 300 python scripts/gpu_narrow_ab.py 32768 0.1 8 4 4 > $OUT/ab_$t.txt 2>&1
  echo "== $t"; grep "run" $OUT/ab_$t.txt | cut -c1-330
done
