#!/bin/bash
# round 5, run z2: whole GPU suite after the partitioned-mode work
cd /root/repo
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/r05z2_gpu_suite_tail.txt
cat gpurun_out/r05z2_gpu_suite_tail.txt
