#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 300 python scripts/r05/handles_probe.py kitti00 dummy_first 2>&1 | grep -v amdgpu.ids | tee $out/r05k_dummy_first.log
GPU_MAX_HW_QUEUES=8 timeout 300 python scripts/r05/handles_probe.py kitti00 pcg_graph=0 2>&1 | grep -v amdgpu.ids | tee $out/r05k_nograph_q8.log
GPU_MAX_HW_QUEUES=8 timeout 300 python scripts/r05/handles_probe.py kitti07 pcg_graph=0 2>&1 | grep -v amdgpu.ids | tee -a $out/r05k_nograph_q8.log
