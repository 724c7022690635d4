#!/bin/bash
# rocprofv3 kernel stats of the linearise + Schur kernels for a list of shapes:  gpurun -- bash scripts/experiments/r03_kstats.sh <tag> kitti00 s2m
tag=$1; shift
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for shape in "$@"; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats_$shape -- python $root/scripts/prof_run.py $shape 3 > $out/${tag}_stats_$shape.log 2>&1
  f=$(find $out/${tag}_stats_$shape -name '*kernel_stats.csv' | head -1)
  echo "== $shape" >> $out/${tag}_kstats.txt
  python - "$f" >> $out/${tag}_kstats.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('lm_pass', 'schur_pass', 'trial_tail', 'reduce_report', 'setup_expand', 'pta_assemble', 'pcg_spmv', 'pcg2_fused')):
        print("%-60s calls %5s avg %9.1f us" % (n[:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
cat $out/${tag}_kstats.txt
