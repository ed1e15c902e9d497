#!/bin/bash
# rocprofv3 passes for profiles/ (run on the GPU box via gpurun).  $1 = tag, rest = bench args
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --no-cpu $@"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES -d $OUT/pmc1 -o pmc1 -- $BENCH > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc2 -- $BENCH > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- $BENCH > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc4 -- $BENCH > $OUT/pmc4.log 2>&1
cd $OUT && find . -name "*.csv" | head -40; du -sh .
for f in $(find . -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f; done
tail -2 trace.log
