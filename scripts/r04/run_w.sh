#!/bin/bash
# aggregate size at the large shapes after the cheaper sweep and the faster SpMV
out=gpurun_out; mkdir -p $out
for a in 24 28 32 36; do timeout 300 python scripts/kernel_times.py s2m pcg_aggregate=$a 2>&1 | grep -v amdgpu.ids; done | tee $out/r04w_aggregate_resweep.txt
for a in 40 48 56 64; do timeout 300 python scripts/kernel_times.py g4m pcg_aggregate=$a 2>&1 | grep -v amdgpu.ids; done | tee -a $out/r04w_aggregate_resweep.txt
for a in 12 14 16 18; do timeout 300 python scripts/kernel_times.py kitti00 pcg_aggregate=$a 2>&1 | grep -v amdgpu.ids; done | tee -a $out/r04w_aggregate_resweep.txt
