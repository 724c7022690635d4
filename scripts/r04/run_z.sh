#!/bin/bash
# large-shape SpMV with non-temporal matrix loads (the stream should not evict the coarse inverse / vectors from the L2s), full runs, same box
out=gpurun_out; mkdir -p $out
D=cuda-bundle-adjustment_amd/csrc
for rep in 1 2; do for lib in $D/libcuba_hip.so $D/libexp_spmv_nt.so; do for s in s2m g4m; do CUBA_HIP_LIB_F64=$lib timeout 300 python scripts/kernel_times.py $s 2>&1 | grep -v amdgpu.ids; done; done; done | tee $out/r04z_spmv_nontemporal.txt
