#!/bin/bash
# GPU call: rocprofv3 evidence for the exact reduced solve: kernel stats of a forced KITTI-00 run, and matrix-core busy cycles of the
# trailing update at S2M size (PMC pass on its own, kernel-trace only beside it)
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r05w_direct_stats -- python $GRAFT_REPO_ROOT/scripts/r05/direct_probe.py kitti00 > $out/r05w_direct_stats.log 2>&1
f=$(find $out/r05w_direct_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $out/r05w_exact_solve_kitti00_kernel_stats.csv && head -8 "$f" | cut -c1-160
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/r05w_direct_pmc -- python $GRAFT_REPO_ROOT/scripts/r05/direct_large.py s2m > $out/r05w_direct_pmc.log 2>&1
tail -3 $out/r05w_direct_pmc.log | cut -c1-200
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for f in glob.glob(out + "/r05w_direct_pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("cubahip::", "").replace("void ", "")
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
with open(out + "/r05w_exact_solve_s2m_mfma_pmc.txt", "w") as fh:
    for k in sorted(tot):
        if "chol" in k:
            m, g = tot[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0), tot[k].get("GRBM_GUI_ACTIVE", 0)
            line = f"{k}: launches {cnt[(k, 'GRBM_GUI_ACTIVE')]}, SQ_VALU_MFMA_BUSY_CYCLES {m:.4g}, GRBM_GUI_ACTIVE {g:.4g}, ratio {m / g if g else 0:.2f}"
            print(line); fh.write(line + "\n")
PY
rm -rf $out/r05w_direct_pmc $out/r05w_direct_stats
