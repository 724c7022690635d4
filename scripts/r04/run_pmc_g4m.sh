#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, kernel-trace only) of a G4M run: are the SpMV's matrix bytes HBM bytes?
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/r04g4m_pmc_$c -- python $root/scripts/prof_run.py g4m 0 > $out/r04g4m_pmc_$c.log 2>&1 </dev/null
done
cd $root
python scripts/pmc_traffic.py $out/r04g4m_pmc_FETCH_SIZE $out/r04g4m_pmc_WRITE_SIZE | sed 's#prof_run.py kitti00 1#prof_run.py g4m 0#' > $out/r04_g4m_pmc_traffic.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_g4m_pmc_traffic.json'))
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['hbm_bytes_raw']*kv[1]['launches'])[:8]:
    print(k[:50], v['launches'], 'raw MB %.1f  fetch-x2 MB %.1f'%(v['hbm_bytes_raw']/1e6, v['hbm_bytes_fetch_x2']/1e6))
PY
