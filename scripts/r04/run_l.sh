#!/bin/bash
out=gpurun_out; mkdir -p $out
( for s in kitti00 s2m g4m; do timeout 300 python scripts/kernel_times.py $s; done ) > $out/r04l_kernel_times.txt 2>&1
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q ) > $out/r04l_parity.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -k "baseline_shape or duplicate or setup_agree or partition" ) > $out/r04l_configs.log 2>&1
grep -v amdgpu $out/r04l_kernel_times.txt | cut -c1-330; tail -2 $out/r04l_parity.log; tail -2 $out/r04l_configs.log
