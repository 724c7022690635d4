#!/bin/bash
# GPU call: device-resident LM decision -- bit-identity test, then the whole suite and timings
out=gpurun_out; mkdir -p $out
( timeout 300 python -m pytest tests -q -x -s -m gpu -k "device_resident_lm" ) 2>&1 | grep -v "^$" | tail -12 | cut -c1-300
( time timeout 1500 python -m pytest tests -q -m gpu ) > $out/r05o_gpu_suite.log 2>&1
grep -v "^$" $out/r05o_gpu_suite.log | grep "^FAILED\|^ERROR\|passed\|failed" | cut -c1-300
for s in kitti07 kitti00 s2m g4m; do for o in device_lm_decision=0 device_lm_decision=1; do timeout 300 python scripts/r05/shapes_time.py $s $o 2>&1 | grep -v amdgpu.ids | cut -c1-140; done; done | tee $out/r05o_device_decision_ab.log
