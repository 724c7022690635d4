"""round 6: N exact reduced solves on one shape (for rocprofv3 --kernel-trace --stats)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from cuba_amd.capi import HipSolver
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_named
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
shape, n = sys.argv[1], int(sys.argv[2])
opts = {k: float(v) for k, v in (a.split("=") for a in sys.argv[3:])}
fp = flatten(synth_named(shape))
b = HipSolver(fp, RK, reduced_solver=1, **opts); lam = 1e-5 * b.max_diagonal(); b.set_lambda(lam)
ts = []
for _ in range(n):
    t = time.perf_counter(); assert b.solve(); b.array("xp")[:1]; ts.append(time.perf_counter() - t)
print(f"{shape} {opts}: schur + exact solve + back-substitution wall {1e3 * min(ts):.3f} ms (min of {n})", flush=True)
