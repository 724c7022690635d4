#!/bin/bash
# A/B of the Schur block pass: camera-frame form (default), the same with loads one trip ahead, and the Jacobian form of rounds 1-2.
#   gpurun -- bash scripts/experiments/r03_block_pass_form.sh
out=gpurun_out/r03x_block_pass_form.txt
mkdir -p gpurun_out; : > $out
for shape in kitti00 kitti07 s2m; do
  for v in "" CUBA_HIP_BLOCK_PASS_PIPELINED=1 CUBA_HIP_BLOCK_PASS_JACOBIANS=1; do
    echo "== $shape ${v:-camera-frame form}" >> $out
    env $v timeout 300 python scripts/kernel_times.py $shape >> $out 2>&1
  done
done
echo "== mixed precision, kitti00" >> $out
for v in "" CUBA_HIP_BLOCK_PASS_PIPELINED=1 CUBA_HIP_BLOCK_PASS_JACOBIANS=1; do
  env $v timeout 300 python scripts/kernel_times.py kitti00 mixed_precision=1 >> $out 2>&1
done
cat $out
