#!/bin/bash
# GPU call: staged Schur block pass -- parity tests, then A/B on all shapes
out=gpurun_out; mkdir -p $out
( timeout 600 python -m pytest tests -q -x -m gpu -k "staged_block_pass" ) 2>&1 | grep -v "^$" | tail -15 | cut -c1-400
for s in kitti00 kitti07 s2m g4m; do for o in schur_staged=0 schur_staged=1; do CUBA_HIP_DEBUG=1 timeout 300 python scripts/r05/shapes_time.py $s $o 2>&1 | grep "staged block pass\|^$s" | sort -u | cut -c1-330; done; done | tee $out/r05p_staged_ab.log
