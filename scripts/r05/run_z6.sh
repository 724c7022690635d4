#!/bin/bash
# round 5, run z6: the GPU suite twice in a row on one box (flakiness soak), exactly as the driver runs it (-x)
cd /root/repo
for i in 1 2; do
  timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
done | tee gpurun_out/r05z6_gpu_suite_twice.txt
