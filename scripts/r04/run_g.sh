#!/bin/bash
out=gpurun_out; mkdir -p $out
python - <<'PY'
import sys; sys.path.insert(0, ".")
from cuba_amd.synth import synth_named
synth_named("kitti00").to_json("/tmp/k00.json")
PY
exe=./cuda-bundle-adjustment_amd/host/samples/sample_ba_from_file
CUBA_HIP_DEBUG=1 CUBA_HIP_NO_STRUCTURE_CACHE=1 $exe /tmp/k00.json 10 1 2>&1 | grep "structure\|optimize:\|cuba host\]\|set_graph" > $out/r04g_debug_new.txt
CUBA_HIP_DEBUG=1 $exe /tmp/k00.json 10 1 2>&1 | grep "structure\|optimize:\|cuba host\]\|set_graph" > $out/r04g_debug_same.txt
cat $out/r04g_debug_new.txt $out/r04g_debug_same.txt | cut -c1-260
