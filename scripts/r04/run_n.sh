#!/bin/bash
out=gpurun_out; mkdir -p $out
(
for a in 28 36 40; do timeout 300 python scripts/kernel_times.py s2m pcg_aggregate=$a; done
for a in 48 64 72; do timeout 300 python scripts/kernel_times.py g4m pcg_aggregate=$a; done
for a in 6 5; do timeout 200 python scripts/kernel_times.py kitti07 pcg_aggregate=$a; done
timeout 200 python scripts/kernel_times.py kitti07
) > $out/r04n_sweeps.txt 2>&1
grep -v amdgpu $out/r04n_sweeps.txt | cut -c1-150
