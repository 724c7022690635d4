#!/bin/bash
# round 5, run z10: the contract-wall leg alone, twice (r05fin4 measured 9.4 ms on its box where r05fin2 / r05fin3 had 8.1 / 8.5)
cd /root/repo
for i in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-shapes --no-concurrent --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['ms_per_step'], json.dumps(b['contract_wall']['unchanged_topology']), json.dumps(b['contract_wall']['new_topology']))"
done | tee gpurun_out/r05z10_contract_wall_again.txt
