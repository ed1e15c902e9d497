#!/bin/bash
# Round-end check on the GPU box: the whole GPU suite; if green, the rocprofv3 passes at the bench's batch and width, the
# traffic entry stamped with this build's kernel_sha, and bench.py.  Everything lands under gpurun_out/final_r02/.
cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r02; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -4 $O/pytest.log
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" $O/pytest.log | head -30; exit 1; fi
bash scripts/gpu_profile_r02.sh 32768 2 > $O/profile.log 2>&1
python scripts/update_traffic.py gpurun_out/prof_r02_final 32768 2 | tee $O/traffic.log
cp profiles/r02_traffic.json $O/r02_traffic.json
grep "run2" gpurun_out/prof_r02_final/trace.log | cut -c1-220
python bench.py > $O/bench.json.log 2> $O/bench.err; tail -c 2500 $O/bench.json.log | head -c 1200; echo; tail -2 $O/bench.err
