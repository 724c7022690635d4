#!/bin/bash
# round 6: kernel split of the exact solve (rocprofv3 kernel stats)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out
for shape in kitti00 g4m; do
  rm -rf /tmp/prof_$shape
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$shape -- python scripts/r06/exact_only.py $shape 20 > $out/r06b_${shape}_log.txt 2>&1
  f=$(find /tmp/prof_$shape -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $out/r06b_exact_solve_${shape}_kernel_stats.csv && head -12 "$f" | cut -c1-200
  tail -2 $out/r06b_${shape}_log.txt
done
