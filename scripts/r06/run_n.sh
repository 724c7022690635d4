#!/bin/bash
# A/B over the two-level kernel's workgroup size for coarse dimensions up to 768 (256 / 320 / 384 = default / 448 / 512 threads): wall time of a
# 10-iteration run + rocprofv3 kernel averages.  (First pass, all size classes at once: 384 / 512 / 768 / 1024 threads -> 6.86 / 6.99 / 7.18 / 10.7 ms.)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06n_fused_wg_ab.txt
for rep in 1 2; do
for lib in libcuba_hip.so libcuba_hip_wg256.so libcuba_hip_wg320.so libcuba_hip_wg448.so libcuba_hip_wg512.so; do
  for shape in kitti00 kitti07; do
    python $R/scripts/r06/fused_wg_ab.py $lib $shape 20 2>/dev/null
  done
done
done > $OUT
for lib in libcuba_hip.so libcuba_hip_wg512.so; do
  rm -rf /tmp/prof_n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_n -- python $R/scripts/r06/fused_wg_ab.py $lib kitti00 5 > /dev/null 2>&1
  f=$(find /tmp/prof_n -name "*kernel_stats.csv" | head -1)
  echo "== $lib (rocprofv3 kernel averages, ns)" >> $OUT
  python - "$f" >> $OUT <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "pcg2_fused" in r["Name"] or "pcg_spmv" in r["Name"]:
        print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"])
PY
done
cat $OUT
