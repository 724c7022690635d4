#!/bin/bash
out=gpurun_out; mkdir -p $out
D=cuda-bundle-adjustment_amd/csrc
for lib in $D/libcuba_hip.so $D/libexp_spmv_fewer_parts_timing.so; do for s in s2m g4m; do CUBA_HIP_LIB_F64=$lib timeout 200 python scripts/r04/spmv_timing_only.py $s 2>&1 | grep -v amdgpu.ids | tail -2; done; done | tee $out/r04t_spmv_fewer_parts_timing.txt
