#!/bin/bash
# rocprofv3 counter passes of the set-full leg (run on the GPU box via gpurun): two SQ passes, FETCH_SIZE and WRITE_SIZE in passes of
# their own (MI355X_MICROARCH.md's recipe, as scripts/gpu_profile_r06.sh for the batch path).  $1 = tag of the output directory
TAG=${1:-r06_setfull}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --leg set_full"
run() { tag=$1; shift; timeout -k 5 240 rocprofv3 "$@" --output-format csv -d $OUT/$tag -o p -- $CMD > $OUT/$tag.log 2>&1 < /dev/null; }
run pmc1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES
run pmc2 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE
run pmc3 --pmc FETCH_SIZE
run pmc4 --pmc WRITE_SIZE
python $GRAFT_REPO_ROOT/scripts/summarize_pmc_csv.py $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 > $OUT/pmc_summary.txt 2>&1
rm -rf $OUT/pmc*/*.csv 2>/dev/null
grep -i "setfull" $OUT/pmc_summary.txt | head -40
