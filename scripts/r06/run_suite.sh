#!/bin/bash
# round 6: the whole GPU suite as the driver runs it (+ durations), tail into gpurun_out/
mkdir -p gpurun_out
tag=${1:-r06_suite}
timeout 3300 python -m pytest tests/ -q -m gpu --durations=15 2>&1 | tail -60 > gpurun_out/${tag}_tail.txt
cat gpurun_out/${tag}_tail.txt
