# the narrow search kernel built under other LLVM scheduling strategies (scripts/build_variant.sh sched_* -mllvm -amdgpu-sched-strategy=...)
OUT=gpurun_out/r06_ac
mkdir -p $OUT
for v in default sched_ilp sched_mem sched_iter; do
  if [ $v = default ]; then timeout 300 python scripts/gpu_narrow_ab.py 32768 0.1 8 4 3 > $OUT/$v.txt 2>&1
  else TBC_LIB_PATH=$GRAFT_REPO_ROOT/jepsen-tigerbeetle_amd/csrc/variants/libtbcheck_$v.so timeout 300 python scripts/gpu_narrow_ab.py 32768 0.1 8 4 3 > $OUT/$v.txt 2>&1; fi
  echo "== $v"; grep "run" $OUT/$v.txt | cut -c1-260
done
