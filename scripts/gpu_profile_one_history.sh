#!/bin/bash
# rocprofv3 counter passes of ONE 10k-op history through tbc_check (scripts/gpu_one_history.py): the level sweep's instruction mix
TAG=${1:-r06_one}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/scripts/gpu_one_history.py competition 1"
run() { tag=$1; shift; (cd $GRAFT_REPO_ROOT && timeout -k 5 240 rocprofv3 "$@" --output-format csv -d $OUT/$tag -o p -- python scripts/gpu_one_history.py competition 1 > $OUT/$tag.log 2>&1 < /dev/null); }
run trace --kernel-trace --stats
run pmc1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES
run pmc2 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE
run pmc3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_BRANCH
python $GRAFT_REPO_ROOT/scripts/summarize_pmc_csv.py $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 > $OUT/pmc_summary.txt 2>&1
f=$(ls $OUT/trace/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv
rm -rf $OUT/pmc*/*.csv $OUT/trace 2>/dev/null
cat $OUT/pmc_summary.txt | cut -c1-60,58-200 | grep -i "sweep\|pack_one" | head -60; head -8 $OUT/kernel_stats.csv | cut -c1-200; tail -3 $OUT/pmc3.log
