#!/bin/bash
out=gpurun_out; mkdir -p $out
python - <<'PY'
import sys; sys.path.insert(0, ".")
from cuba_amd.synth import synth_named
synth_named("kitti00").to_json("/tmp/k00.json")
PY
exe=./cuda-bundle-adjustment_amd/host/samples/sample_ba_from_file
for i in 1 2 3; do $exe /tmp/k00.json 10 1 | grep "BA total"; done > $out/r04e_wall.txt 2>&1
for i in 1 2 3; do CUBA_HIP_NO_STRUCTURE_CACHE=1 $exe /tmp/k00.json 10 1 | grep "BA total"; done >> $out/r04e_wall.txt 2>&1
CUBA_HIP_DEBUG=1 $exe /tmp/k00.json 10 1 > $out/r04e_debug_same.txt 2>&1
CUBA_HIP_DEBUG=1 CUBA_HIP_NO_STRUCTURE_CACHE=1 $exe /tmp/k00.json 10 1 > $out/r04e_debug_new.txt 2>&1
cat $out/r04e_wall.txt
