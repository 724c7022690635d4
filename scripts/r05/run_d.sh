#!/bin/bash
# fourth GPU call of round 5: the whole GPU suite on the tree with the exact fallback on by default, then the G4M reference pin
# (verdict item 2: CUBA_TEST_REF_G4M=1, never run before), then the queue-count dependence of the two-chain micro-benchmark
out=gpurun_out; mkdir -p $out
( time timeout 1500 python -m pytest tests -q -m gpu -x ) > $out/r05d_gpu_suite.log 2>&1
tail -6 $out/r05d_gpu_suite.log | cut -c1-400
( time CUBA_TEST_REF_G4M=1 timeout 1500 python -m pytest tests/test_ref_lm.py -q -x -s -k g4m_full ) > $out/r05d_g4m_ref_pin.log 2>&1
grep "g4m_full\|passed\|failed\|real" $out/r05d_g4m_ref_pin.log | cut -c1-1200
for q in 2 8 16; do echo "GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q timeout 100 scripts/ubench/two_chains 2>&1 | grep "grid  64, 5 us kernels, hipGraph\|mixed"; done | tee $out/r05d_two_chains_queues.log
