#!/bin/bash
# A/B of the pieces of the block pass: whole-wave threshold, 64-byte inverse rows
out=gpurun_out/${1:-r03z2}_block_pass_ab.txt; : > $out
for shape in kitti00 s2m; do
for v in "" CUBA_HIP_BP_HEAVY=1000000 CUBA_HIP_BP_HEAVY=128 CUBA_HIP_BP_HEAVY=256 CUBA_HIP_BLOCK_PASS_INV_FROM_LM_SYS=1 "CUBA_HIP_BP_HEAVY=1000000 CUBA_HIP_BLOCK_PASS_INV_FROM_LM_SYS=1"; do
  echo "== $shape ${v:-default}" >> $out
  env $v timeout 200 python scripts/kernel_times.py $shape 2>&1 | grep -o "linearize_schur [0-9.]* us" >> $out
done; done
echo "== kitti00 mixed" >> $out
for v in "" CUBA_HIP_BP_HEAVY=1000000 CUBA_HIP_BLOCK_PASS_INV_FROM_LM_SYS=1; do
  echo "-- ${v:-default}" >> $out
  env $v timeout 200 python scripts/kernel_times.py kitti00 mixed_precision=1 2>&1 | grep -o "linearize_schur [0-9.]* us" >> $out
done
cat $out
