#!/usr/bin/env python3
"""Stage timestamps of the two PCG kernels (last active launch of a solve), from the libcuba_hip_trace.so build.

    make -C cuda-bundle-adjustment_amd/csrc libcuba_hip_trace.so
    gpurun -- python scripts/trace_pcg.py [shape]
Prints, per kernel, the distribution over waves of each stage's time since the first wave started (100 MHz clock).
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CUBA_HIP_LIB_F64"] = os.path.join(ROOT, "cuda-bundle-adjustment_amd", "csrc", "libcuba_hip_trace.so")
import numpy as np  # noqa: E402
from cuba_amd import capi  # noqa: E402
from cuba_amd.graph import flatten  # noqa: E402
from cuba_amd.synth import synth_named  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "kitti00"
fp = flatten(synth_named(shape))
rk = ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815))))
opts = {k: float(v) for k, v in (a.split("=") for a in sys.argv[2:] if "=" in a)}      # e.g. spmv_upper=1
h = capi.HipSolver(fp, rk, **opts)
h.build_structure()
h.optimize(3)
upper = opts.get("spmv_upper", -1) == 1 or (opts.get("spmv_upper", -1) < 0 and fp.Pf > 1536)
lib = capi.load_library("f64")
buf = np.zeros((3, 8192, 8), dtype=np.uint64)
rc = lib.cuba_hip_debug_read_trace(buf.ctypes.data_as(C.c_void_p))          # the PCG translation unit's stamps (kernels 0, 1)
assert rc == 0, rc
gj = np.zeros((3, 8192, 8), dtype=np.uint64)
rc = lib.cuba_hip_debug_read_trace_coarse(gj.ctypes.data_as(C.c_void_p))    # the coarse translation unit's (kernel 2)
assert rc == 0, rc
if upper:
    # the upper-triangle iteration: kernel 0 = pcg_spmv_upper, 1 = the preconditioner-only instantiation of the fused kernel, 2 = pcg_rows
    table = ((0, "pcg_spmv_upper", ["entry", "scalars+row pointers", "operands + products", "reductions + stores"]),
             (2, "pcg_rows", ["entry", "loads landed (q rows complete)", "barrier (alpha)", "end"]),
             (1, "pcg2_fused<PRE>", ["entry", "loads landed", "restricted sums", "barrier 1", "barrier yc", "end"]))
else:
    buf[2] = gj[2]
    table = None
for kid, name, stages in table or ((0, "pcg_spmv", ["entry", "indices+scalars", "operands", "fold+barrier", "end"]),
                          (1, "pcg2_fused", ["entry", "loads landed", "restricted sums", "barrier 1", "barrier yc", "end"]),
                          (2, "dense_gj_step (last launch that has a look-ahead workgroup... the last step has none)", ["entry", "tile loads", "tiles done", "end (look-ahead chain in one workgroup)"])):
    t = buf[kid].astype(np.int64)
    if kid == 2 and not upper:
        la = t[8000:8004]
        print("dense_gj_step, look-ahead workgroup (4 waves): ns since entry at [loads, tiles done, chain done]:",
              [[int((w[s] - w[0]) * 10) for s in (1, 2, 3)] for w in la if w[0] > 0])
        t = t[:8000]
    on = t[:, 0] > 0
    t = t[on]
    n = len(stages)
    t0 = t[:, 0].min()
    print(f"{name}: {on.sum()} waves, span {(t[:, n - 1].max() - t0) * 10} ns")
    for s in range(n):
        d = (t[:, s] - t0) * 10
        print(f"  {stages[s]:18s} min {d.min():6d}  p50 {int(np.median(d)):6d}  p90 {int(np.percentile(d, 90)):6d}  max {d.max():6d} ns")
    d = (t[:, 0] - t[:, 7]) * 10
    print(f"  wave start -> entry mark (first kernel-argument fetch): p10 {int(np.percentile(d, 10))} p50 {int(np.median(d))} p90 {int(np.percentile(d, 90))} ns; "
          f"first wave start -> last wave start {(t[:, 7].max() - t[:, 7].min()) * 10} ns")
    dur = (t[:, n - 1] - t[:, 0]) * 10
    print(f"  per-wave duration  min {dur.min()} p50 {int(np.median(dur))} max {dur.max()} ns")
    for s in range(1, n):
        d = (t[:, s] - t[:, s - 1]) * 10
        print(f"  stage {stages[s]:18s} p10 {int(np.percentile(d, 10)):6d} p50 {int(np.median(d)):6d} p90 {int(np.percentile(d, 90)):6d} max {d.max():6d} ns")
h.close()
