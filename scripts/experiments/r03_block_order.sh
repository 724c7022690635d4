# block-pass scheduling experiment: XCD-aware orders of the off-diagonal blocks (host pipeline), kernel time + HBM-side fetch
mkdir -p gpurun_out; out=$PWD/gpurun_out; root=$PWD
for m in 0 1 2; do
  if [ $m = 0 ]; then unset CUBA_HIP_BLOCK_ORDER_XCD; else export CUBA_HIP_BLOCK_ORDER_XCD=$m; fi
  echo "== CUBA_HIP_BLOCK_ORDER_XCD=$m" >> $out/r03j_block_order.txt
  (timeout 120 python scripts/kernel_times.py kitti00 device_setup=0 2>&1 | tail -1) >> $out/r03j_block_order.txt
  (cd /tmp && export TMPDIR=/tmp CUBA_PROF_OPTS=device_setup=0 && timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/r03j_pmc_$m -- python $root/scripts/prof_run.py kitti00 1 > $out/r03j_pmc_$m.log 2>&1)
  python - <<PY >> $out/r03j_block_order.txt
import csv, glob
tot, cnt = {}, {}
for f in glob.glob("$out/r03j_pmc_$m/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row.get("Counter_Name") != "FETCH_SIZE": continue
        n = row["Kernel_Name"].split("(")[0].replace("void cubahip::", "")
        tot[n] = tot.get(n, 0) + float(row["Counter_Value"]); cnt[n] = cnt.get(n, 0) + 1
for n in sorted(tot):
    if "schur" in n or "lm_pass" in n or "block_pass" in n or "pose_pass" in n:
        print("   FETCH_SIZE %-50s %8.1f MB per launch (raw KiB counter / 1024, %d launches)" % (n[:50], tot[n] / cnt[n] / 1024, cnt[n]))
PY
done
cat $out/r03j_block_order.txt
