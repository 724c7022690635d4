#!/bin/bash
# seventh GPU call of round 5: upper-triangle iteration v2 (two steps in flight, 16 lower neighbours per trip), per-kernel times
out=gpurun_out; mkdir -p $out
( timeout 300 python -m pytest tests -q -x -m gpu -k "upper_triangle" ) 2>&1 | tail -3
for s in s2m g4m; do for o in spmv_upper=0 spmv_upper=1; do timeout 300 python scripts/r05/shapes_time.py $s $o 2>&1 | grep -v amdgpu.ids; done; done | tee $out/r05g_upper_ab.log
timeout 300 python scripts/r05/handles_probe.py kitti00 2>&1 | grep -v amdgpu.ids | tee $out/r05g_handles.log
