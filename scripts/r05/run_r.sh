#!/bin/bash
# GPU call: internal landmark order -- the new test, the whole suite, then all shapes with / without
out=gpurun_out; mkdir -p $out
( timeout 300 python -m pytest tests -q -x -m gpu -k "internal_landmark_order or staged_block" ) 2>&1 | grep -v "^$" | tail -12 | cut -c1-400
( time timeout 1500 python -m pytest tests -q -m gpu ) > $out/r05r_gpu_suite.log 2>&1
grep -v "^$" $out/r05r_gpu_suite.log | grep "^FAILED\|^ERROR\|passed\|failed" | cut -c1-300
for s in kitti00 kitti07 s2m g4m; do for o in "landmark_reorder=0 schur_staged=0" "landmark_reorder=1 schur_staged=0" "landmark_reorder=1 schur_staged=1"; do CUBA_HIP_DEBUG=1 timeout 300 python scripts/r05/shapes_time.py $s $o 2>&1 | grep "staged block pass\|^$s" | sort -u | cut -c1-330; done; done | tee $out/r05r_landmark_order_ab.log
