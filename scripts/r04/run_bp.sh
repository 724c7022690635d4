#!/bin/bash
# Schur block pass: product-list entries fetched one trip ahead -- same-box A/B + the tests that pin the reduced system
out=gpurun_out; mkdir -p $out
D=cuda-bundle-adjustment_amd/csrc
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "stage_parity or schur_paths or golden or more_than_64 or kitti07" 2>&1 | tail -2
for lib in $D/libexp_before.so $D/libcuba_hip.so $D/libexp_before.so $D/libcuba_hip.so; do for s in kitti00 g4m; do CUBA_HIP_LIB_F64=$lib timeout 300 python scripts/kernel_times.py $s 2>&1 | grep -v amdgpu.ids; done; done | tee $out/r04bp_index_prefetch.txt
