#!/bin/bash
# host side of the contract wall: threads of the host pool x items per thread in the pointer-chasing loops of initialize() / write-back
python - <<'PY'
import sys; sys.path.insert(0, ".")
from cuba_amd.synth import synth_named
synth_named("kitti00").to_json("/tmp/k00.json")
PY
EXE=./cuda-bundle-adjustment_amd/host/samples/sample_ba_from_file
for cfg in "4 20000" "8 20000" "12 20000" "16 20000" "24 20000" "32 20000" "16 40000" "8 40000" "16 20000" "32 20000"; do
  set -- $cfg
  for rep in 1 2 3 4 5; do
    CUBA_HIP_HOST_THREADS=$1 CUBA_HOST_GRAIN=$2 CUBA_HIP_DEBUG=1 $EXE /tmp/k00.json 10 1 2>&1 | grep -E "BA total|initialize\(\)|write-back|get_solution|create \+ set_graph" | tail -5 | tr '\n' ' ' | sed "s/^/threads $1 grain $2: /; s/\[cuba host\]//g; s/  */ /g"
    echo
  done
done > gpurun_out/r06u_host_threads_ab.txt 2>&1
cat gpurun_out/r06u_host_threads_ab.txt
