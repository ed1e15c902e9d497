# A/B runs of the wide search kernel on the GPU box: width (configs per round) on both bench workloads
cd $GRAFT_REPO_ROOT
timeout 120 python scripts/gpu_quick_bench.py 32768 10000 1 8 0.1 1 2 2>&1 | grep "run1"
for w in 4 2 1; do timeout 200 python scripts/gpu_quick_bench.py 1024 10000 $w 256 0.5 1 1 2>&1 | grep "run0"; done
for w in 4 2; do timeout 200 python scripts/gpu_quick_bench.py 4096 10000 $w 32 0.3 1 2 2>&1 | grep "run1"; done
