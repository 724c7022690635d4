#!/bin/bash
# first GPU call of round 5: the exact reduced solve -- kernel tests, per-shape timing, the Tukey start, then the tests that changed
out=gpurun_out; mkdir -p $out
( time timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "exact_reduced or dense_inverse" ) > $out/r05a_kernel_tests.log 2>&1
tail -15 $out/r05a_kernel_tests.log
for s in kitti07 kitti00; do CUBA_HIP_DEBUG=1 timeout 300 python scripts/r05/direct_probe.py $s 2>&1 | grep "exact reduced solve\|^$s" | tail -8; done > $out/r05a_direct_times.log 2>&1
cat $out/r05a_direct_times.log
( time timeout 600 python scripts/r05/direct_probe.py tukey ) > $out/r05a_tukey.log 2>&1
cut -c1-600 $out/r05a_tukey.log
( time timeout 900 python -m pytest tests -q -x -m gpu -k "pcg_max_iter or rejected_trials" ) > $out/r05a_tests.log 2>&1
tail -30 $out/r05a_tests.log
