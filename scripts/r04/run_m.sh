#!/bin/bash
out=gpurun_out; mkdir -p $out
C=cuda-bundle-adjustment_amd/csrc
(
for a in 4 6 8 10; do timeout 200 python scripts/kernel_times.py kitti07 pcg_aggregate=$a; done
for lib in default libexp_period_1024_2048.so libexp_period_256_512.so; do
  for s in kitti00 s2m; do
    if [ $lib = default ]; then timeout 300 python scripts/kernel_times.py $s; else CUBA_HIP_LIB_F64=$PWD/$C/$lib timeout 300 python scripts/kernel_times.py $s; fi
  done
done
CUBA_HIP_LIB_F64=$PWD/$C/libexp_period_1024_4096.so timeout 300 python scripts/kernel_times.py g4m
timeout 300 python scripts/kernel_times.py g4m
) > $out/r04m_sweeps.txt 2>&1
grep -v amdgpu $out/r04m_sweeps.txt | cut -c1-150
