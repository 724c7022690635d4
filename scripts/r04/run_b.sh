#!/bin/bash
out=gpurun_out; mkdir -p $out
( time timeout 600 python scripts/r04/diag_b.py tukey ) > $out/r04b_tukey.log 2>&1
for s in kitti07 kitti00 s2m g4m; do CUBA_HIP_DEBUG=1 timeout 300 python scripts/r04/diag_b.py $s 2>&1 | grep "PCG:\|^$s" > $out/r04b_rz0_$s.log; done
( time timeout 1500 python -m pytest tests -q -m gpu -x --deselect "tests/test_ref_lm.py::test_full_size_rejected_trials_follow_the_reference[k00_lm10m_tukey]" ) > $out/r04b_gpu_suite.log 2>&1
tail -5 $out/r04b_gpu_suite.log; cat $out/r04b_tukey.log | cut -c1-400
