#!/bin/bash
# round 6: kernel split of a batched run (4 x KITTI-00, 8 x KITTI-07)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out
for cfg in "kitti00 4" "kitti07 8"; do
  set -- $cfg
  rm -rf /tmp/prof_b_$1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b_$1 -- python scripts/r06/batch_only.py $1 $2 10 > $out/r06h_batch_$1_log.txt 2>&1
  f=$(find /tmp/prof_b_$1 -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $out/r06h_batch_$1_kernel_stats.csv && head -14 "$f" | cut -c1-150
  grep "batch wall\|solo" $out/r06h_batch_$1_log.txt
done
