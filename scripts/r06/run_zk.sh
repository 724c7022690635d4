#!/bin/bash
# randomised edit sequences through the C++ API (host/samples/edit_fuzz.cpp): long-lived object against fresh objects, bit for bit
python - <<'PY'
import sys; sys.path.insert(0, ".")
from cuba_amd.synth import synth_ba
synth_ba(120, 6000, 24000, seed=9).to_json("/tmp/g120.json")
synth_ba(400, 12000, 50000, seed=3).to_json("/tmp/g400.json")
synth_ba(1800, 40000, 160000, seed=5).to_json("/tmp/g1800.json")
PY
EXE=./cuda-bundle-adjustment_amd/host/samples/edit_fuzz
for cfg in "/tmp/g120.json 60 1" "/tmp/g120.json 60 2" "/tmp/g400.json 40 3" "/tmp/g1800.json 25 4"; do
  set -- $cfg
  CUBA_HIP_HEURISTICS=0 timeout 600 $EXE $1 $2 $3 > /tmp/ef.txt 2>&1; echo "== $cfg: $(tail -1 /tmp/ef.txt)"; grep -B1 -A1 "DIFFERENT" /tmp/ef.txt | head -20
done > gpurun_out/r06zk_edit_fuzz.txt 2>&1
cat gpurun_out/r06zk_edit_fuzz.txt; tail -12 /tmp/ef.txt
