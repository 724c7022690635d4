#!/bin/bash
# tenth GPU call: same box, default vs pcg_graph=0 handles side by side
out=gpurun_out; mkdir -p $out
timeout 300 python scripts/r05/handles_probe.py kitti00 2>&1 | grep -v amdgpu.ids | tee $out/r05j_handles_default.log
timeout 300 python scripts/r05/handles_probe.py kitti00 pcg_graph=0 2>&1 | grep -v amdgpu.ids | tee $out/r05j_handles_nograph.log
timeout 300 python scripts/r05/handles_probe.py kitti00 2>&1 | grep -v amdgpu.ids | tee -a $out/r05j_handles_default.log
