"""A/B of solver options in one process: min and median wall of a 10-iteration run over several repeats.
   python scripts/experiments/option_ab.py kitti00 spin_wait=0 spin_wait=1,speculate_tail=0 ..."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from cuba_amd.capi import HipSolver
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
shape = sys.argv[1]
fp = flatten(synth_named(shape))
variants = [dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in a.split(",") if kv) for a in sys.argv[2:]] or [{}]
hs = []
for v in variants:
    h = HipSolver(fp, RK, **v); h.build_structure(); hs.append(h)
q0, t0, X0 = hs[0].state()
for h in hs: h.optimize(10)
times = [[] for _ in hs]
for rep in range(12):
    for i, h in enumerate(hs):
        h.set_state(q0, t0, X0)
        t = time.perf_counter(); got = h.optimize(10)["chi2"]; times[i].append(time.perf_counter() - t)
for v, ts, h in zip(variants, times, hs):
    c = h.counters()
    print("%-40s min %.3f ms  median %.3f ms  looks/run %.1f" % (v, min(ts) * 1e3, np.median(ts) * 1e3, c["pcg_host_looks"] / 13.0), flush=True)
