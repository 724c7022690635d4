#!/bin/bash
# Gauss-Jordan step kernel, one tile per workgroup, four LDS arrays: inverse tests + kernel times
out=gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "dense_inverse or coarse or preconditioner or golden" 2>&1 | tail -2
for s in kitti00 s2m g4m kitti07; do timeout 300 python scripts/kernel_times.py $s 2>&1 | grep -v amdgpu.ids; done | tee $out/r04r_gj_one_tile_per_workgroup.txt
