#!/bin/bash
# GPU call: whole suite on the tree with the upper-triangle iteration automatic + graphs policy, then the bench line with the new legs
out=gpurun_out; mkdir -p $out
( time timeout 1500 python -m pytest tests -q -m gpu ) > $out/r05m_gpu_suite.log 2>&1
grep -v "^$" $out/r05m_gpu_suite.log | grep "^FAILED\|^ERROR\|passed\|failed" | cut -c1-300
( time timeout 900 python bench.py --steps 50 --warmup 10 ) > $out/r05m_bench.json 2> $out/r05m_bench.err
tail -c 1500 $out/r05m_bench.json; tail -5 $out/r05m_bench.err
