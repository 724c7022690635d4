#!/bin/bash
# round 5, run z7: the GPU suite as the driver runs it; the summary line goes to gpurun_out/
cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r05z7_gpu_suite_full.txt 2>&1
grep -E "passed|failed|error" gpurun_out/r05z7_gpu_suite_full.txt | tail -5
