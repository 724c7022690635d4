#!/bin/bash
# GPU call: does landmark order matter?  benchmark graphs with landmark ids in creation order (first observing pose), both block passes
out=gpurun_out; mkdir -p $out
for s in kitti00 g4m; do for o in schur_staged=0 schur_staged=1; do CUBA_HIP_DEBUG=1 timeout 300 python scripts/r05/shapes_time.py $s $o sortlm 2>&1 | grep "staged block pass\|^$s" | sort -u | cut -c1-330; done; done | tee $out/r05q_sorted_landmarks.log
