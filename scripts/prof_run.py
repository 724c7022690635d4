"""Workload for rocprofv3: KITTI-00-shaped graph, one warm-up LM run + N timed runs of 10 iterations."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from cuba_amd.capi import HipSolver

name = sys.argv[1] if len(sys.argv) > 1 else "kitti00"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
fp = flatten(synth_named(name))
opts = dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in os.environ.get("CUBA_PROF_OPTS", "").split(",") if kv)   # e.g. CUBA_PROF_OPTS=device_setup=0
h = HipSolver(fp, RK, **opts)
h.build_structure()
q0, t0, X0 = h.state()
for r in range(runs + 1):
    h.set_state(q0, t0, X0)
    t = time.time(); res = h.optimize(10); dt = time.time() - t
    print("run", r, "%.1f ms" % (dt * 1e3), "chi2[-1]", res["chi2"][-1], h.counters())
h.close()
sys.stdout.flush()
