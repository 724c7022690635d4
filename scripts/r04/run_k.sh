#!/bin/bash
out=gpurun_out; mkdir -p $out
bash scripts/r04/run_g.sh > /dev/null
bash scripts/r04/run_f.sh > $out/r04k_f.log 2>&1
( for a in 14 12 18 20; do timeout 300 python scripts/kernel_times.py kitti00 pcg_aggregate=$a; done ) > $out/r04k_agg.txt 2>&1
grep "structure\|set_graph:" $out/r04g_debug_new.txt | tail -9 | cut -c1-110
cat $out/r04f_wall.txt; grep -E "passed|failed" $out/r04f_parity.log; cut -c1-200 $out/r04k_agg.txt | grep kitti00
