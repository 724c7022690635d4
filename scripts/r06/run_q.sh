#!/bin/bash
# two-level kernel of the small size class (coarse dimension <= 512): 256-thread workgroups (default, two per compute unit) against 512 (one)
mkdir -p gpurun_out
for v in default wg512; do
  if [ $v = wg512 ]; then export CUBA_HIP_LIB_F64=$PWD/cuda-bundle-adjustment_amd/csrc/libcuba_hip_wg512.so; fi
  for rep in 1 2; do
  python bench.py --concurrent-child --shape kitti00 --concurrent-counts 2,4 2>/dev/null | grep CONCURRENT > /tmp/c00.txt
  python bench.py --concurrent-child --shape kitti07 --concurrent-counts 4,8 2>/dev/null | grep CONCURRENT > /tmp/c07.txt
  python - $v <<'PY'
import json, sys
for shp, f in (("kitti00", "/tmp/c00.txt"), ("kitti07", "/tmp/c07.txt")):
    d = json.loads(open(f).read()[len("CONCURRENT "):])
    print(sys.argv[1], shp, "solo %.2f ms" % d["solo_wall_ms_10iter"])
    for n, g in d["batch"].items(): print("  batch  ", n, "x%.2f" % g["throughput_vs_one_graph"], "%.2f ms for all" % g["wall_ms_10iter_all_graphs"], g["batched_reduced_solves"], g["bit_identical_to_solo"])
PY
  done
done > gpurun_out/r06q_small_class_wg_ab.txt 2>&1
cat gpurun_out/r06q_small_class_wg_ab.txt
