"""One leg of the A/B over the two-level kernel's workgroup size: python fused_wg_ab.py <library> [shape] [runs]
(libraries: make -C cuda-bundle-adjustment_amd/csrc libcuba_hip_wg384.so libcuba_hip_wg768.so libcuba_hip_wg1024.so)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["CUBA_HIP_LIB_F64"] = os.path.join(ROOT, "cuda-bundle-adjustment_amd", "csrc", sys.argv[1])
import numpy as np
from cuba_amd import capi
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_named
shape = sys.argv[2] if len(sys.argv) > 2 else "kitti00"
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 20
fp = flatten(synth_named(shape))
rk = ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815))))
h = capi.HipSolver(fp, rk)
h.snapshot_state(1)
ms = []
for i in range(runs + 3):
    h.restore_state(1)
    t0 = time.perf_counter(); r = h.optimize(10); t1 = time.perf_counter()
    if i >= 3: ms.append((t1 - t0) * 1e3)
c = h.counters()
print(f"{sys.argv[1]:24s} {shape}: min {min(ms):.3f} ms  median {np.median(ms):.3f} ms  pcg iterations {c['pcg_iterations'] // (runs + 3)} per run  chi2 {r['chi2'][-1]:.9e}")
