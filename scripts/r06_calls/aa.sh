# what the front walk's stores cost: variants that leave out the front records (1), the lists and twin masks (2), the lookahead records (4), all (7)
OUT=gpurun_out/r06_aa
mkdir -p $OUT
timeout 300 python scripts/gpu_walk_exp.py > $OUT/default.txt 2>&1; tail -3 $OUT/default.txt
for v in 1 2 4 7; do
  TBC_LIB_PATH=$GRAFT_REPO_ROOT/jepsen-tigerbeetle_amd/csrc/variants/libtbcheck_walkexp$v.so timeout 120 python scripts/gpu_walk_exp.py > $OUT/walkexp$v.txt 2>&1; tail -3 $OUT/walkexp$v.txt | cut -c1-300
done
