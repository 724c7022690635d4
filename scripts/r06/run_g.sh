#!/bin/bash
# round 6: several graphs on one GPU -- threads vs one batched launch chain (default configuration: no hipGraphs)
mkdir -p gpurun_out
python bench.py --concurrent-child --shape kitti00 --concurrent-counts 2,4 2>/dev/null | grep CONCURRENT > gpurun_out/r06g_concurrent_kitti00.txt
python bench.py --concurrent-child --shape kitti07 --concurrent-counts 4,8 2>/dev/null | grep CONCURRENT > gpurun_out/r06g_concurrent_kitti07.txt
python - <<'PY'
import json
for shp in ("kitti00", "kitti07"):
    d = json.loads(open(f"gpurun_out/r06g_concurrent_{shp}.txt").read()[len("CONCURRENT "):])
    print(shp, "solo %.2f ms" % d["solo_wall_ms_10iter"])
    for n, g in d["groups"].items(): print("  threads", n, "x%.2f" % g["throughput_vs_one_graph"], ["%.2f" % v for v in g["per_graph_wall_ms_10iter"]], g["bit_identical_to_solo"])
    for n, g in d["batch"].items(): print("  batch  ", n, "x%.2f" % g["throughput_vs_one_graph"], "%.2f ms for all" % g["wall_ms_10iter_all_graphs"], g["batched_reduced_solves"], g["bit_identical_to_solo"])
PY
