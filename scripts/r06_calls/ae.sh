# (the -DTBC_WALK_XCD variant of pack_open.hip this call compares was an experiment's and is gone: profiles/NOTES_r06.md)
# the front walk with every XCD given a contiguous eighth of the (history, chunk) pairs (build variant -DTBC_WALK_XCD)
OUT=gpurun_out/r06_ag; mkdir -p $OUT
for v in default walkxcd default walkxcd; do
  LIB=$GRAFT_REPO_ROOT/jepsen-tigerbeetle_amd/csrc/variants/libtbcheck_$v.so; [ $v = default ] && LIB=$GRAFT_REPO_ROOT/jepsen-tigerbeetle_amd/csrc/libtbcheck.so
  TBC_LIB_PATH=$LIB timeout 300 python scripts/gpu_narrow_ab.py 32768 0.1 8 4 3 > $OUT/$v.txt 2>&1
  echo "== $v"; grep "run" $OUT/$v.txt | cut -c1-200
done
