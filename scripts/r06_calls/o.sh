mkdir -p gpurun_out/r06_o
for v in mw3 mw2; do
  if [ $v = default ]; then unset TBC_LIB_PATH; else export TBC_LIB_PATH=$PWD/jepsen-tigerbeetle_amd/csrc/variants/libtbcheck_$v.so; fi
  echo "== $v" >> gpurun_out/r06_o/min_waves_low.txt
  timeout 600 python scripts/gpu_narrow_ab.py 32768 0.1 8 4 3 2>&1 | grep -v amdgpu.ids | tail -4 >> gpurun_out/r06_o/min_waves_low.txt
done
cat gpurun_out/r06_o/min_waves_low.txt | cut -c1-260
