#!/bin/bash
# third GPU call of round 5: why do concurrent handles serialise?  micro-benchmark of dependent chains on several streams, eager vs graph,
# two processes vs two threads
out=gpurun_out; mkdir -p $out
timeout 300 scripts/ubench/two_chains 2>&1 | tee $out/r05c_two_chains.log
timeout 200 python scripts/r05/handles_probe.py kitti00 pcg_graph=0 2>&1 | grep -v amdgpu.ids | tee $out/r05c_handles_eager.log
( timeout 100 python scripts/r05/handles_probe.py kitti00 loop 2>&1 | grep loop ) | tee $out/r05c_one_process.log
( timeout 100 python scripts/r05/handles_probe.py kitti00 loop 2>&1 | grep loop & timeout 100 python scripts/r05/handles_probe.py kitti00 loop 2>&1 | grep loop; wait ) | tee $out/r05c_two_processes.log
( time timeout 600 python -m pytest tests -q -x -m gpu -k "rejected_trials" ) > $out/r05c_tests.log 2>&1; tail -4 $out/r05c_tests.log | cut -c1-800
