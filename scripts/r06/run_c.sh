#!/bin/bash
# round 6: GPU tests around the sparse exact solve
mkdir -p gpurun_out
timeout 3000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "exact or upper_triangle or pcg_max_iter or dense_inverse" -s 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r06c_tests.txt
timeout 3000 python -m pytest tests/test_ref_lm.py -m gpu -x -q -k "tukey" -s 2>&1 | grep -v "^$" | tail -25 >> gpurun_out/r06c_tests.txt
cat gpurun_out/r06c_tests.txt
