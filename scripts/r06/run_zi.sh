#!/bin/bash
# kernel stats + L2-fabric traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the batched iteration kernels: 4 x KITTI-00, 8 x KITTI-07 on the final tree
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out
for cfg in "kitti00 4" "kitti07 8"; do
  set -- $cfg
  rm -rf /tmp/prof_zi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_zi -- python scripts/r06/batch_only.py $1 $2 10 > $out/r06zi_batch_$1_log.txt 2>&1
  f=$(find /tmp/prof_zi -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $out/r06zi_batch_$1_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/prof_zi_$c
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_zi_$c -- python scripts/r06/batch_only.py $1 $2 2 > /dev/null 2>&1
  done
  python - $1 $2 <<'PY' >> $out/r06zi_batch_traffic.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/prof_zi_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            key = next((k for k in ("pcg_spmv_batch_kernel", "pcg2_fused_batch_kernel", "pcg_spmv_kernel", "pcg2_fused_kernel", "schur_pass_batch_kernel", "lm_pass_batch_kernel", "trial_tail_batch_kernel") if k in n), None)
            if key is None: continue
            a = acc[key][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print(f"{sys.argv[1]} x {sys.argv[2]}: mean per dispatch, KiB as reported (FETCH_SIZE under-counts wide coalesced streams by 2 x on gfx950: scripts/pmc_traffic.py)")
for k in acc:
    fs = acc[k]["FETCH_SIZE"]; ws = acc[k]["WRITE_SIZE"]
    f = fs[0] / max(fs[1], 1); w = ws[0] / max(ws[1], 1)
    print(f"  {k:28s} launches {fs[1]:6d}  FETCH_SIZE {f:10.1f} KiB  WRITE_SIZE {w:10.1f} KiB")
PY
done
cat $out/r06zi_batch_traffic.txt
for s in kitti00 kitti07; do python - $out/r06zi_batch_${s}_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]: print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "%8.1f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"][:5])
PY
done
