#!/bin/bash
# large-shape SpMV with 8 block rows per workgroup (half the partial sets the two-level kernel reads), full runs, same box
out=gpurun_out; mkdir -p $out
D=cuda-bundle-adjustment_amd/csrc
for rep in 1 2; do for lib in $D/libcuba_hip.so $D/libexp_spmv_rows8.so; do for s in s2m g4m; do CUBA_HIP_LIB_F64=$lib timeout 300 python scripts/kernel_times.py $s 2>&1 | grep -v amdgpu.ids; done; done; done | tee $out/r04z_spmv_rows8.txt
