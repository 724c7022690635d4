#!/bin/bash
# GPU call: single-sweep panel kernel of the exact solve (tests + time), aggregate sizes under the upper-triangle iteration, mixed precision
out=gpurun_out; mkdir -p $out
( timeout 300 python -m pytest tests -q -x -m gpu -k "exact_reduced or pcg_max_iter or rejected_trials" ) 2>&1 | tail -3
for s in kitti07 kitti00; do CUBA_HIP_DEBUG=1 timeout 300 python scripts/r05/direct_probe.py $s 2>&1 | grep "exact reduced solve\|^$s" | tail -3; done | tee $out/r05n_direct_times.log
for a in 40 48 64 72; do timeout 300 python scripts/r05/shapes_time.py g4m pcg_aggregate=$a 2>&1 | grep -v amdgpu.ids; done | tee $out/r05n_aggregate_sweep.log
for a in 28 32 36 48; do timeout 300 python scripts/r05/shapes_time.py s2m pcg_aggregate=$a 2>&1 | grep -v amdgpu.ids; done | tee -a $out/r05n_aggregate_sweep.log
for s in kitti00 g4m; do timeout 300 python scripts/r05/shapes_time.py $s mixed_precision=1 2>&1 | grep -v amdgpu.ids; done | tee $out/r05n_mixed.log
