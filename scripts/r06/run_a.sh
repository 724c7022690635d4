#!/bin/bash
# round 6, first GPU look at the sparse exact solve
mkdir -p gpurun_out
export CUBA_HIP_DEBUG=1
timeout 1500 python scripts/r06/direct_probe.py kernels kitti07 kitti00 s2m g4m > gpurun_out/r06a_probe.txt 2> gpurun_out/r06a_err.txt
grep "symbolic phase" gpurun_out/r06a_err.txt | sort | uniq -c >> gpurun_out/r06a_probe.txt
grep "exact reduced solve:" gpurun_out/r06a_err.txt | awk '{print $5, $7, $8}' | sort | uniq -c | sort -k2n | head -40 >> gpurun_out/r06a_probe.txt
grep -v "^\[cuba_hip\]" gpurun_out/r06a_err.txt | tail -20 >> gpurun_out/r06a_probe.txt
rm -f gpurun_out/r06a_err.txt
tail -80 gpurun_out/r06a_probe.txt
