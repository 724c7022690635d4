#!/bin/bash
# round 6: matrix-core busy cycles of the sparse exact solve's factorisation launches (PMC pass on its own, kernel-trace only beside it), G4M shape
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/r06j_pmc -- python $GRAFT_REPO_ROOT/scripts/r06/exact_only.py g4m 5 > $out/r06j_pmc.log 2>&1
tail -2 $out/r06j_pmc.log | cut -c1-200
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for f in glob.glob("/tmp/r06j_pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("cubahip::", "").replace("void ", "")
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
with open(out + "/r06j_exact_solve_g4m_mfma_pmc.txt", "w") as fh:
    fh.write("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over scripts/r06/exact_only.py g4m 5 (sums over all launches; GRBM_GUI_ACTIVE is per XCD-sum as reported)\n")
    for k in sorted(tot):
        if "schol" in k:
            m, g = tot[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0), tot[k].get("GRBM_GUI_ACTIVE", 0)
            line = f"{k}: launches {cnt[(k, 'GRBM_GUI_ACTIVE')]}, SQ_VALU_MFMA_BUSY_CYCLES {m:.4g}, GRBM_GUI_ACTIVE {g:.4g}, ratio {m / g if g else 0:.3f}"
            print(line); fh.write(line + "\n")
PY
rm -rf /tmp/r06j_pmc
