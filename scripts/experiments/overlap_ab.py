"""A/B of the overlapped coarse inversion (second stream) against the lagged in-stream refresh.  python scripts/experiments/overlap_ab.py [shape]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from cuba_amd.capi import HipSolver
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
shape = sys.argv[1] if len(sys.argv) > 1 else "kitti00"
fp = flatten(synth_named(shape))
for ov in (0, 1, 0, 1):
    h = HipSolver(fp, RK, coarse_overlap=ov); h.build_structure(); q0, t0, X0 = h.state()
    h.optimize(10)
    best = 1e9
    for rep in range(4):
        h.set_state(q0, t0, X0); c0 = h.counters()
        t = time.perf_counter(); got = h.optimize(10)["chi2"]; best = min(best, time.perf_counter() - t)
    c = h.counters()
    print("overlap %d iters %5d refreshes %d wall %.2f ms chi2 %.9e" % (ov, c["pcg_iterations"] - c0["pcg_iterations"], c["coarse_refreshes"] - c0["coarse_refreshes"], best * 1e3, got[-1]), flush=True)
    h.close()
