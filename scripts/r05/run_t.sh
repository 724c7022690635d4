#!/bin/bash
# GPU call: waves per workgroup of the upper-triangle SpMV (experiment builds libexp_upper_w{2,8}.so against the default 4)
out=gpurun_out; mkdir -p $out
for s in s2m g4m; do
  timeout 300 python scripts/r05/shapes_time.py $s 2>&1 | grep "^$s" | cut -c1-300
  for w in 2 8; do CUBA_HIP_LIB_F64=$PWD/cuda-bundle-adjustment_amd/csrc/libexp_upper_w$w.so timeout 300 python scripts/r05/shapes_time.py $s 2>&1 | grep "^$s" | sed "s/^/[waves $w] /" | cut -c1-300; done
done | tee $out/r05t_upper_waves.log
