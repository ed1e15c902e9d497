#!/bin/bash
# rocprofv3 passes for profiles/ (run on the GPU box via gpurun): kernel trace + stats and two SQ counter passes (FETCH_SIZE /
# WRITE_SIZE in passes of their own when $4 = traffic) on the batch path.  $1 = batch (default 32768), $2 = lanes per history
# (8 / 16 / 32, or 64 = one history per wavefront), $3 = tag of the output directory, $5 = visited-set entries per op (default 4, the bench's).
B=${1:-32768}
L=${2:-8}
TAG=${3:-r06_l$L}
VPO=${5:-4}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/scripts/gpu_narrow_ab.py $B 0.1 $L $VPO 3"
run() { tag=$1; shift; timeout -k 5 240 rocprofv3 "$@" --output-format csv -d $OUT/$tag -o p -- $CMD > $OUT/$tag.log 2>&1 < /dev/null; }
# kernel trace + stats of the bench's own timed region (two batches in flight: the search kernel's average duration here is the one
# bench.py reports as roofline.kernel_ms); the counter passes below run ONE batch alone (counts and bytes do not depend on company)
BENCH="python $GRAFT_REPO_ROOT/bench.py --only-headline --batch $B --lanes $( [ $L = 8 ] && echo 0 || echo $L ) --visited-per-op $VPO"
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- $BENCH > $OUT/bench_under_rocprof.log 2>&1 < /dev/null
CMD1="python $GRAFT_REPO_ROOT/scripts/gpu_narrow_ab.py $B 0.1 $L $VPO 2"
timeout -k 5 240 $CMD1 > $OUT/trace.log 2>&1 < /dev/null
run pmc1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES
run pmc2 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE
if [ "$4" = traffic ]; then
  run pmc3 --pmc FETCH_SIZE
  run pmc4 --pmc WRITE_SIZE
fi
python $GRAFT_REPO_ROOT/scripts/summarize_pmc_csv.py $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 > $OUT/pmc_summary.txt 2>&1
if [ "$4" = traffic ]; then
  mkdir -p $GRAFT_REPO_ROOT/gpurun_out/profiles
  (cd $GRAFT_REPO_ROOT && TBC_TRAFFIC_FILE=r06_traffic.json TBC_TRAFFIC_SOURCE="scripts/gpu_profile_r06.sh; summary committed as profiles/r06_pmc_final.txt" python scripts/update_traffic.py $OUT $B $L $VPO && cp profiles/r06_traffic.json gpurun_out/profiles/) 2>&1 | tail -2
fi
f=$(ls $OUT/trace/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -14 "$f" > $OUT/kernel_stats_head.csv
rm -rf $OUT/trace/*kernel_trace.csv $OUT/pmc*/*.csv 2>/dev/null
tail -2 $OUT/bench_under_rocprof.log | cut -c1-1500; tail -3 $OUT/trace.log; cat $OUT/kernel_stats_head.csv; grep -i "narrow\|beam" $OUT/pmc_summary.txt | head -40
