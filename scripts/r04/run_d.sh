#!/bin/bash
out=gpurun_out; mkdir -p $out
( time timeout 1800 python -m pytest tests -q -m gpu -x ) > $out/r04d_gpu_suite.log 2>&1
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $out/r04d_bench.json 2> $out/r04d_bench.err
tail -6 $out/r04d_gpu_suite.log | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04d_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","value_replay","pcg_iterations","coarse_inline_inversions")}, d["heuristics_off"]["value"], d["contract_wall"])
print({k:(v.get("wall_ms_10iter"),v.get("wall_ms_10iter_replay")) for k,v in d["shapes"].items()})
PY
