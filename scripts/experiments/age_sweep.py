"""Coarse-inverse reuse sweep: iterations / wall per 10-iteration run vs (coarse_max_age, coarse_refresh_growth).  python scripts/experiments/age_sweep.py [shape]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from cuba_amd.capi import HipSolver
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
shape = sys.argv[1] if len(sys.argv) > 1 else "kitti00"
fp = flatten(synth_named(shape))
for age, growth in ((0, 1.25), (2, 1.25), (2, 1e9), (3, 1e9), (4, 1e9), (6, 1e9), (9, 1e9), (4, 2.0), (9, 2.0), (9, 3.0)):
    h = HipSolver(fp, RK, coarse_max_age=age, coarse_refresh_growth=growth); h.build_structure(); q0, t0, X0 = h.state()
    h.optimize(10)
    best = 1e9
    for rep in range(3):
        h.set_state(q0, t0, X0); c0 = h.counters()
        t = time.perf_counter(); got = h.optimize(10)["chi2"]; best = min(best, time.perf_counter() - t)
    print("age %d growth %g iters %5d wall %.2f ms chi2 %.9e" % (age, growth, h.counters()["pcg_iterations"] - c0["pcg_iterations"], best * 1e3, got[-1]), flush=True)
    h.close()
