#!/bin/bash
# host side of the contract wall: threads / grain of the graph re-read in initialize()
python - <<'PY'
import sys; sys.path.insert(0, ".")
from cuba_amd.synth import synth_named
synth_named("kitti00").to_json("/tmp/k00.json")
PY
out=gpurun_out/r03z8_host_threads.txt; : > $out
for cfg in "32 20000" "32 8000" "32 4000" "64 4000" "64 2000" "128 2000" "16 4000"; do
  set -- $cfg
  echo "== threads cap $1, grain $2" >> $out
  for i in 1 2 3; do
    CUBA_HIP_HOST_THREADS=$1 CUBA_HIP_HOST_GRAIN=$2 CUBA_HIP_DEBUG=1 ./cuda-bundle-adjustment_amd/host/samples/sample_ba_from_file /tmp/k00.json 10 1 2>&1 | grep -E "initialize: (landmarks|edges)|initialize\(\)|BA total" | tail -4 | tr '\n' ' ' >> $out; echo >> $out
  done
  for i in 1 2 3; do CUBA_HIP_HOST_THREADS=$1 CUBA_HIP_HOST_GRAIN=$2 ./cuda-bundle-adjustment_amd/host/samples/sample_ba_from_file /tmp/k00.json 10 1 | grep "BA total" | tr '\n' ' ' >> $out; done; echo >> $out
done
cat $out
