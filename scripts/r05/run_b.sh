#!/bin/bash
# second GPU call of round 5: retuned hand-over budget (tests), kernel split of the exact solve, several live handles
out=gpurun_out; mkdir -p $out
( time timeout 900 python -m pytest tests -q -x -m gpu -k "pcg_max_iter or rejected_trials or exact_reduced" ) > $out/r05b_tests.log 2>&1
tail -12 $out/r05b_tests.log | cut -c1-1500
( timeout 300 python scripts/r05/direct_probe.py tukey ) 2>&1 | cut -c1-330 > $out/r05b_tukey.log; cat $out/r05b_tukey.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/r05b_prof -o direct -- python $GRAFT_REPO_ROOT/scripts/r05/direct_probe.py kitti00 > $GRAFT_REPO_ROOT/$out/r05b_prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $out/r05b_prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-200
for s in kitti07 kitti00; do timeout 300 python scripts/r05/handles_probe.py $s; done 2>&1 | grep -v amdgpu.ids | tee $out/r05b_handles.log
GPU_MAX_HW_QUEUES=8 timeout 300 python scripts/r05/handles_probe.py kitti07 2>&1 | grep -v amdgpu.ids | tee $out/r05b_handles_q8.log
