#!/bin/bash
out=gpurun_out; mkdir -p $out
for s in kitti00 kitti07; do timeout 300 python scripts/r05/handles_probe.py $s dummy_first 2>&1 | grep -v amdgpu.ids; done | tee $out/r05l_shared_side.log
for s in kitti00 kitti07; do GPU_MAX_HW_QUEUES=8 timeout 300 python scripts/r05/handles_probe.py $s dummy_first 2>&1 | grep -v amdgpu.ids; done | tee $out/r05l_shared_side_q8.log
