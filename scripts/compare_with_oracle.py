#!/usr/bin/env python3
"""Counterpart of the reference's samples/sample_comparison_with_g2o.cpp with the CPU oracle in g2o's place:
same graph into the exact-solve CPU LM (oracle/ba_oracle.cpp, test infrastructure) and into the HIP path, one
warm-up LM iteration each (ref :303-307), 10 timed iterations each (ref :67-79), then the chi2 table (ref :89-110)
and the RMSE of the estimates (ref :112-136) in the README's format.

    python scripts/compare_with_oracle.py kitti00            # a BASELINE shape (synthetic)
    python scripts/compare_with_oracle.py path/to/graph.json # the reference's JSON schema
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from cuba_amd.capi import HipSolver  # noqa: E402
from cuba_amd.graph import Graph, flatten, write_back  # noqa: E402
from cuba_amd.synth import SHAPES, synth_named  # noqa: E402
from oracle.oracle import OracleSolver  # noqa: E402

RK = ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815))))   # Huber, ref :195-200
arg = sys.argv[1] if len(sys.argv) > 1 else "kitti07"
g_cpu = synth_named(arg) if arg in SHAPES else Graph.from_json(arg)
import copy  # noqa: E402
g_gpu = copy.deepcopy(g_cpu)
print(f"=== Graph size :\nnum poses      : {g_cpu.nposes}\nnum landmarks  : {g_cpu.nlandmarks}\nnum edges      : {g_cpu.nedges}\n")

# warm-up: initialize + optimize(1) on both, results written back into the graphs (it moves the estimates)
fp = flatten(g_cpu); o = OracleSolver(fp, RK); o.optimize(1); write_back(g_cpu, fp, *o.state())
fp = flatten(g_gpu); h = HipSolver(fp, RK); h.optimize(1); write_back(g_gpu, fp, *h.state()); h.close()

t0 = time.perf_counter(); fp_c = flatten(g_cpu); o = OracleSolver(fp_c, RK); rc = o.optimize(10)["chi2"]; t1 = time.perf_counter()
t2 = time.perf_counter(); fp_g = flatten(g_gpu); h = HipSolver(fp_g, RK); rg = h.optimize(10)["chi2"]; t3 = time.perf_counter()
print("=== Processing time : ")
print(f"CPU : {t1 - t0:7.2f} [sec]   (oracle, 1 thread)\nGPU : {t3 - t2:7.2f} [sec]   (flatten + upload + structure + optimize(10))\n")
print("=== Objective function value : ")
print("%10s|%14s|%14s" % ("iteration", "chi2 CPU", "chi2 GPU"))
for i in range(max(len(rc), len(rg))):
    print("%10d|%14s|%14s" % (i + 1, "%.1f" % rc[i] if i < len(rc) else "N/A", "%.1f" % rg[i] if i < len(rg) else "N/A"))
qc, tc, Xc = o.state(); qg, tg, Xg = h.state()
print("\n=== RMSE between CPU estimates and GPU estimates : ")
print("Rotation    : %.2e" % np.sqrt(((qc - qg) ** 2).sum(1).mean()))
print("Translation : %.2e" % np.sqrt(((tc - tg) ** 2).sum(1).mean()))
print("Landmark    : %.2e" % np.sqrt(((Xc - Xg) ** 2).sum(1).mean()))
print("max relative chi2 difference : %.2e" % np.max(np.abs(rc[:len(rg)] - rg[:len(rc)]) / rc[:len(rg)]))
