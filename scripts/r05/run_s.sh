#!/bin/bash
out=gpurun_out; mkdir -p $out
( time timeout 1500 python -m pytest tests -q -m gpu ) > $out/r05s_gpu_suite.log 2>&1
grep -v "^$" $out/r05s_gpu_suite.log | grep "^FAILED\|^ERROR\|passed\|failed" | cut -c1-300
