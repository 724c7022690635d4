import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from cuba_amd.capi import HipSolver
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_named
RK = ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815))))
fp = flatten(synth_named("kitti07"))
for mode in (0, 1):
    h = HipSolver(fp, RK, device_lm_decision=mode); h.optimize(1); h.snapshot_state(); h.optimize(10)
    ts = []
    for _ in range(5):
        h.restore_state(); t = time.perf_counter(); h.optimize(10); ts.append(time.perf_counter() - t)
    print("MODE", mode, "median %.3f ms" % (1e3 * np.median(ts)), flush=True)
    os.environ["CUBA_HIP_DEBUG"] = "1"
    h.restore_state(); h.optimize(10)
    del os.environ["CUBA_HIP_DEBUG"]
    h.close()
