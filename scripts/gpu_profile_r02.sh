#!/bin/bash
# rocprofv3 passes for profiles/ (run on the GPU box via gpurun): kernel trace + stats, two SQ counter passes, FETCH_SIZE and
# WRITE_SIZE in passes of their own (MI355X_MICROARCH.md: HBM / rocprofv3 section), on the batch path at the bench's batch
# size and width.  $1 = batch (default 32768), $2 = search width (default 2: what the bench's workload resolves to),
# $3 = "calib" to also run the known-byte calibration microbenchmark under the same two counters.
B=${1:-32768}
W=${2:-2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r02_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/scripts/gpu_quick_bench.py $B 10000 $W 8 0.1 1"
run() { tag=$1; shift; timeout 400 rocprofv3 "$@" --output-format csv -d $OUT/$tag -o p -- $CMD > $OUT/$tag.log 2>&1 < /dev/null; }
run trace --kernel-trace --stats
run pmc1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES
run pmc2 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE
run pmc3 --pmc FETCH_SIZE
run pmc4 --pmc WRITE_SIZE
if [ "$3" = calib ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/hbm_calib $GRAFT_REPO_ROOT/scripts/hbm_calib.hip
  /tmp/hbm_calib > $OUT/calib_plain.log 2>&1
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -o p -- /tmp/hbm_calib > $OUT/calib_fetch.log 2>&1 < /dev/null
  timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -o p -- /tmp/hbm_calib > $OUT/calib_write.log 2>&1 < /dev/null
fi
python $GRAFT_REPO_ROOT/scripts/summarize_pmc_csv.py $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 > $OUT/pmc_summary.txt 2>&1
f=$(ls $OUT/trace/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -12 "$f" > $OUT/kernel_stats_head.csv
tail -2 $OUT/trace.log; cat $OUT/pmc_summary.txt | grep -i "beam\|open_\|pack" | head -60
