#!/bin/bash
# Kernel times of the four bench shapes (A/B of builds: run it on each).
#   gpurun -- bash scripts/experiments/r03_block_pass_form.sh <tag>
out=gpurun_out/${1:-r03y}_kernel_times.txt
mkdir -p gpurun_out; : > $out
for shape in kitti00 kitti07 s2m g4m; do
  timeout 300 python scripts/kernel_times.py $shape 2>&1 | grep -v amdgpu.ids >> $out
done
timeout 300 python scripts/kernel_times.py kitti00 mixed_precision=1 2>&1 | grep -v amdgpu.ids >> $out
CUBA_HIP_SEPARATE_SCHUR_PASSES=1 timeout 300 python scripts/kernel_times.py kitti00 2>&1 | grep -v amdgpu.ids | sed 's/^/separate passes: /' >> $out
cat $out
