#!/bin/bash
# eighth GPU call of round 5: upper-triangle iteration v3 (transposed products in consumer order) + stage traces
out=gpurun_out; mkdir -p $out
( timeout 300 python -m pytest tests -q -x -m gpu -k "upper_triangle" ) 2>&1 | tail -3
for s in s2m g4m; do timeout 300 python scripts/r05/shapes_time.py $s spmv_upper=1 2>&1 | grep -v amdgpu.ids; done | tee $out/r05h_upper.log
for s in s2m g4m; do timeout 300 python scripts/trace_pcg.py $s spmv_upper=1 2>&1 | grep -v amdgpu.ids; done | tee $out/r05h_trace.log
