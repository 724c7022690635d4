#!/bin/bash
# README protocol on the KITTI-00-shaped graph through the C++ API: warm-up initialize()+optimize(1), then timed initialize()+optimize(10)
set -e
cd "$(dirname "$0")/.."
python - <<'PY'
import sys; sys.path.insert(0, ".")
from cuba_amd.synth import synth_named
synth_named("kitti00").to_json("/tmp/k00.json")
PY
./cuda-bundle-adjustment_amd/host/samples/sample_ba_from_file /tmp/k00.json 10 1 | grep "BA total"
CUBA_HIP_DEBUG=1 CUBA_HIP_PROFILE=1 ./cuda-bundle-adjustment_amd/host/samples/sample_ba_from_file /tmp/k00.json 10 1 | grep -E "BA total|msec|cuba_hip"
