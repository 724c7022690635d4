#!/bin/bash
out=gpurun_out; mkdir -p $out
( timeout 600 python -m pytest tests -q -x -m gpu -k "pcg_max_iter or rejected_trials or exact_reduced" ) 2>&1 | tail -3
for s in kitti07 kitti00; do CUBA_HIP_DEBUG=1 timeout 300 python scripts/r05/direct_probe.py $s 2>&1 | grep "exact reduced solve\|^$s" | tail -3; done | tee $out/r05v_direct_lookahead.log
timeout 300 python scripts/r05/direct_probe.py tukey 2>&1 | grep "^default\|^tol" | cut -c1-200 | tee -a $out/r05v_direct_lookahead.log
timeout 600 python scripts/r05/direct_large.py s2m g4m 2>&1 | grep -v amdgpu.ids | tee -a $out/r05v_direct_lookahead.log
