#!/bin/bash
# PMC counters of the Schur pass (and the PCG kernels for comparison): what the texture addresser / L1 / L2 see.
#   gpurun -- bash scripts/experiments/r03_pmc_schur.sh r03
tag=${1:-r03}; root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TAGRAM0_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/${tag}_pmcs_$i -- python $root/scripts/prof_run.py kitti00 1 > $out/${tag}_pmcs_$i.log 2>&1
done
cd $root
python - "$out" "$tag" > $out/${tag}_pmc_schur.txt <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(f"{out}/{tag}_pmcs_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = next((k for k in ("schur_pass_kernel", "lm_pass_kernel<1", "pcg_spmv_kernel", "pcg2_fused_kernel", "trial_tail_kernel") if k in n), None)
        if key is None: continue
        a = acc[key][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
names = sorted({c for k in acc for c in acc[k]})
print("%-34s" % "counter (mean per dispatch)" + "".join("%22s" % k for k in acc))
for c in names:
    print("%-34s" % c + "".join("%22.4g" % (acc[k][c][0] / max(acc[k][c][1], 1)) for k in acc))
PY
cat $out/${tag}_pmc_schur.txt
