#!/bin/bash
out=gpurun_out; mkdir -p $out
python - <<'PY'
import sys; sys.path.insert(0, ".")
from cuba_amd.synth import synth_named
synth_named("kitti00").to_json("/tmp/k00.json")
PY
exe=./cuda-bundle-adjustment_amd/host/samples/sample_ba_from_file
( for i in 1 2 3 4 5 6 7; do $exe /tmp/k00.json 10 1 | grep "BA total"; done; echo new; for i in 1 2 3 4 5 6 7; do CUBA_HIP_NO_STRUCTURE_CACHE=1 $exe /tmp/k00.json 10 1 | grep "BA total"; done ) > $out/r04f_wall.txt 2>&1
CUBA_HIP_DEBUG=1 CUBA_HIP_NO_STRUCTURE_CACHE=1 $exe /tmp/k00.json 10 1 2>&1 | grep -v "PCG:" > $out/r04f_debug_new.txt
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q ) > $out/r04f_parity.log 2>&1
( time timeout 600 python bench.py --steps 20 --warmup 5 --no-shapes --no-cpu-baseline ) > $out/r04f_bench.json 2> $out/r04f_bench.err
cat $out/r04f_wall.txt; tail -4 $out/r04f_parity.log
