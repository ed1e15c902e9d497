#!/bin/bash
# memory-system counters of the batch path (run on the GPU box via gpurun): where a load's time goes -- L1 (TCP) accesses and
# miss latency, address translation (UTCL1), L2 (TCC) hits / misses / fabric requests, the texture addresser's stalls.
# $1 = batch, $2 = lanes per history, $3 = tag
B=${1:-32768}
L=${2:-8}
TAG=${3:-r03_mem_l$L}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/scripts/gpu_narrow_ab.py $B 0.1 $L 8 2"
run() { tag=$1; shift; timeout -k 5 90 rocprofv3 "$@" --output-format csv -d $OUT/$tag -o p -- $CMD > $OUT/$tag.log 2>&1 < /dev/null; }
# (at most four TCP / TCC counters a pass: more "exceeds the capabilities of the hardware" and rocprofv3 then hangs until killed)
run m1 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum
run m2 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum
if [ "$4" = ta ]; then
  run m4 --pmc TA_TA_BUSY_sum TA_FLAT_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
fi
if [ "$4" = more ]; then
  run m3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
  run m4 --pmc TA_TA_BUSY_sum TA_FLAT_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
fi
python $GRAFT_REPO_ROOT/scripts/summarize_pmc_csv.py $OUT/m1 $OUT/m2 $OUT/m3 $OUT/m4 > $OUT/pmc_summary.txt 2>&1
rm -rf $OUT/m*/*.csv 2>/dev/null
grep -i "narrow\|beam_kernel" $OUT/pmc_summary.txt | cut -c1-36,58-140
grep -h "exceeds\|Error\|error" $OUT/m*.log | head -5 | cut -c1-200
