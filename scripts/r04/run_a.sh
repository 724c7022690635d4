#!/bin/bash
# round 4, first GPU call: the new parity tests + the new bench protocol
out=gpurun_out; mkdir -p $out
( time timeout 900 python -m pytest tests/test_ref_lm.py -x -q -s -k "s2m_full or rejected" ) > $out/r04a_ref_lm.log 2>&1
( time timeout 600 python -m pytest tests/test_gpu_configs.py -x -q -s -k "config4" ) > $out/r04a_config4.log 2>&1
( time timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -s -k "snapshot or fp32_coarse" ) > $out/r04a_parity.log 2>&1
( time timeout 900 python -m pytest tests/test_dist.py -x -q -s -k "bench_" ) > $out/r04a_dist.log 2>&1
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $out/r04a_bench.json 2> $out/r04a_bench.err
tail -3 $out/r04a_ref_lm.log $out/r04a_config4.log $out/r04a_parity.log $out/r04a_dist.log
tail -c 1500 $out/r04a_bench.json
