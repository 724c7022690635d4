#!/bin/bash
# Where the Schur block pass spends its time (timing experiments with wrong results; block pass as its own launch).
out=gpurun_out/r03x_block_pass_where2.txt; : > $out
for x in 0 3 4 6 7 8 9 10; do
  echo "== separate passes, experiment $x" >> $out
  CUBA_HIP_SEPARATE_SCHUR_PASSES=1 CUBA_HIP_BLOCK_PASS_EXPERIMENT=$x timeout 200 python scripts/kernel_times.py kitti00 2>&1 | grep -o "linearize_schur [0-9.]* us" >> $out
done
cat $out
