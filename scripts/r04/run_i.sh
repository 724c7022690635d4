#!/bin/bash
out=gpurun_out; mkdir -p $out
( time timeout 1800 python -m pytest tests -q -m gpu -x ) > $out/r04i_gpu_suite.log 2>&1
tail -4 $out/r04i_gpu_suite.log | cut -c1-200
bash scripts/profile_round.sh r04fin 2>&1 | tail -3
