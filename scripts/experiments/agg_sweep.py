"""Aggregate-size sweep of the two-level preconditioner: iterations / wall per 10-iteration run.  python scripts/experiments/agg_sweep.py [shape]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from cuba_amd.capi import HipSolver
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
shape = sys.argv[1] if len(sys.argv) > 1 else "kitti00"
fp = flatten(synth_named(shape))
aggs = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8, 12, 16, 20, 24, 32, 48]
for agg in aggs:
    h = HipSolver(fp, RK, pcg_aggregate=agg); h.build_structure(); q0, t0, X0 = h.state()
    h.optimize(10)
    best = 1e9
    for rep in range(3):
        h.set_state(q0, t0, X0); c0 = h.counters()
        t = time.perf_counter(); got = h.optimize(10)["chi2"]; best = min(best, time.perf_counter() - t)
    print("agg %d iters %5d wall %.2f ms chi2 %.9e" % (agg, h.counters()["pcg_iterations"] - c0["pcg_iterations"], best * 1e3, got[-1]), flush=True)
    h.close()
