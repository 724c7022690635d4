#!/bin/bash
# sixth GPU call of round 5: upper-triangle PCG iteration -- parity tests, then S2M / G4M / KITTI-00 A/B (one handle per process)
out=gpurun_out; mkdir -p $out
( time timeout 600 python -m pytest tests -q -x -m gpu -k "upper_triangle or rejected_trials or shuffled_pose or two_level" ) > $out/r05f_tests.log 2>&1
grep -v "^$" $out/r05f_tests.log | tail -15 | cut -c1-600
for s in s2m g4m kitti00; do for o in spmv_upper=0 spmv_upper=1; do timeout 300 python scripts/r05/shapes_time.py $s $o 2>&1 | grep -v amdgpu.ids; done; done | tee $out/r05f_upper_ab.log
GPU_MAX_HW_QUEUES=8 timeout 300 python scripts/r05/handles_probe.py kitti00 2>&1 | grep -v amdgpu.ids | tee $out/r05f_handles_q8.log
timeout 300 python scripts/r05/handles_probe.py kitti00 2>&1 | grep -v amdgpu.ids | tee $out/r05f_handles.log
