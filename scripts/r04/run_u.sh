#!/bin/bash
# symmetric sweep (upper tiles only): inverse tests, the parity file, kernel times
out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3
for s in kitti00 s2m g4m kitti07; do timeout 300 python scripts/kernel_times.py $s 2>&1 | grep -v amdgpu.ids; done | tee $out/r04u_symmetric_sweep.txt
