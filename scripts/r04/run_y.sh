#!/bin/bash
# rocprofv3 kernel stats of the large shapes (the round's SpMV and sweep changes are aimed at them)
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for s in s2m g4m; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r04y_${s}_stats -- python $root/scripts/prof_run.py $s 3 > $out/r04y_${s}_stats.log 2>&1 </dev/null
  f=$(find $out/r04y_${s}_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $out/r04y_${s}_kernel_stats.csv
  head -8 $out/r04y_${s}_kernel_stats.csv | cut -c1-160
done
