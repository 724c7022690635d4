#!/bin/bash
# large-shape SpMV (one wave per block row): one level of the fixed-width row in flight at occupancy 5 instead of two at 4 -- same-box A/B
out=gpurun_out; mkdir -p $out
D=cuda-bundle-adjustment_amd/csrc
timeout 900 python -m pytest tests/test_gpu_configs.py -q -m gpu -x -k "baseline_shape or block_rows_wider or shuffled" 2>&1 | tail -2
for rep in 1 2; do for lib in $D/libexp_spmv_two_levels.so $D/libcuba_hip.so; do for s in s2m g4m; do CUBA_HIP_LIB_F64=$lib timeout 300 python scripts/kernel_times.py $s 2>&1 | grep -v amdgpu.ids; done; done; done | tee -a $out/r04s_spmv_row_occupancy.txt
