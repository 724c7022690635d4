#!/bin/bash
# round 5, run z9: rocprofv3 kernel stats of the reference's own kernels (oracle/_ref, test infrastructure) on the KITTI-00 shape:
# two runs of initialize() + optimize(10) (tests/ref_kernels_profile_run.py)
root=/root/repo; out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r05z9_ref_stats -- python $root/tests/ref_kernels_profile_run.py kitti00_full > $out/r05z9.log 2>&1
f=$(find $out/r05z9_ref_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $out/r05z9_reference_kernels_kitti00_kernel_stats.csv
rm -rf $out/r05z9_ref_stats
tail -2 $out/r05z9.log | cut -c1-400; head -22 $out/r05z9_reference_kernels_kitti00_kernel_stats.csv | cut -c1-220
