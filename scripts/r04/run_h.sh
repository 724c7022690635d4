#!/bin/bash
out=gpurun_out; mkdir -p $out
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "dense_inverse or preconditioner or coarse" ) > $out/r04h_tests.log 2>&1
( for s in kitti00 s2m; do timeout 300 python scripts/kernel_times.py $s; done ) > $out/r04h_kernel_times.txt 2>&1
tail -3 $out/r04h_tests.log; cat $out/r04h_kernel_times.txt | cut -c1-400
