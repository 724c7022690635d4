#!/bin/bash
# overlapped-inversion period after the symmetric sweep: Nc <= A -> every trial, <= B -> every second, else every third
out=gpurun_out; mkdir -p $out
D=cuda-bundle-adjustment_amd/csrc
for lib in $D/libcuba_hip.so $D/libexp_period_1024_2048.so $D/libexp_period_512_4096.so $D/libexp_period_1024_4096.so $D/libexp_period_4096_4096.so; do for s in kitti00 s2m g4m; do CUBA_HIP_LIB_F64=$lib timeout 300 python scripts/kernel_times.py $s 2>&1 | grep -v amdgpu.ids; done; done | tee $out/r04v_period_sweep.txt
