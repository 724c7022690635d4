#!/bin/bash
out=gpurun_out; mkdir -p $out
for q in "" 8; do for s in kitti00 kitti07; do echo "GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q CUBA_HIP_GRAPHS=0 timeout 300 python scripts/r05/handles_probe.py $s 2>&1 | grep -v amdgpu.ids; done; done | tee $out/r05x_handles_inline_inversion.log
