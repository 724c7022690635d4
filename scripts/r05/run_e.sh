#!/bin/bash
# fifth GPU call of round 5: whole GPU suite without -x (which tests does the default-on exact fallback touch?), price of a last-arriver tail
out=gpurun_out; mkdir -p $out
( time timeout 1500 python -m pytest tests -q -m gpu ) > $out/r05e_gpu_suite.log 2>&1
grep -v "^$" $out/r05e_gpu_suite.log | grep "^FAILED\|^ERROR\|passed\|failed" | cut -c1-300
timeout 100 scripts/ubench/last_arriver 2>&1 | tee $out/r05e_last_arriver.log
