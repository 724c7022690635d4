#!/bin/bash
# Gauss-Jordan step kernel with 4 LDS arrays (4 workgroups per CU): column tiles per workgroup 1 / 2 / 3
out=gpurun_out; mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "dense_inverse" 2>&1 | tail -2
for c in 1 2 3; do for s in kitti00 s2m g4m; do echo "GJ_COLS=$c"; CUBA_HIP_GJ_COLS=$c timeout 300 python scripts/kernel_times.py $s 2>&1 | grep -v amdgpu.ids; done; done | tee $out/r04q_gj_cols.txt
