#!/bin/bash
# rocprofv3 on the pack kernels of one batch (run on the GPU box via gpurun): kernel trace + stats, then two SQ counter passes.
# $1 = batch, $2 = tag of the output directory
B=${1:-32768}
TAG=${2:-r03_pack}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/scripts/gpu_narrow_ab.py $B 0.1 8 4 2"
run() { tag=$1; shift; timeout -k 5 150 rocprofv3 "$@" --output-format csv -d $OUT/$tag -o p -- $CMD > $OUT/$tag.log 2>&1 < /dev/null; }
run trace --kernel-trace --stats
run pmc1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run pmc2 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
python $GRAFT_REPO_ROOT/scripts/summarize_pmc_csv.py $OUT/pmc1 $OUT/pmc2 > $OUT/pmc_summary.txt 2>&1
f=$(ls $OUT/trace/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -12 "$f" > $OUT/kernel_stats_head.csv
rm -rf $OUT/trace/*kernel_trace.csv $OUT/pmc*/*.csv 2>/dev/null
tail -2 $OUT/trace.log; cat $OUT/kernel_stats_head.csv; grep -i "walk\|pack_kernel\|counts\|dprod\|meta" $OUT/pmc_summary.txt | cut -c1-150
