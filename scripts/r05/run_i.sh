#!/bin/bash
# ninth GPU call of round 5: several live handles once no handle keeps hipGraphs while another one is alive
out=gpurun_out; mkdir -p $out
for s in kitti00 kitti07; do timeout 300 python scripts/r05/handles_probe.py $s 2>&1 | grep -v amdgpu.ids; done | tee $out/r05i_handles.log
for s in kitti00 kitti07; do GPU_MAX_HW_QUEUES=8 timeout 300 python scripts/r05/handles_probe.py $s 2>&1 | grep -v amdgpu.ids; done | tee $out/r05i_handles_q8.log
for s in s2m g4m; do timeout 300 python scripts/r05/shapes_time.py $s 2>&1 | grep -v amdgpu.ids; done | tee $out/r05i_upper.log
