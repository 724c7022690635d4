"""round 6: N graphs of one shape as one batch, R timed calls (for rocprofv3 --kernel-trace --stats)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from cuba_amd.capi import HipSolver, optimize_batch
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_named
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
shape, n, R = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
fp = flatten(synth_named(shape))
hs = []
for _ in range(n):
    h = HipSolver(fp, RK); h.optimize(1); h.snapshot_state(); h.optimize(10); hs.append(h)
ts = []
for _ in range(R):
    for h in hs: h.restore_state()
    t = time.perf_counter(); chis, b = optimize_batch(hs, 10); ts.append(time.perf_counter() - t)
print(f"{shape} x {n}: batch wall {1e3 * np.median(ts):.2f} ms (median of {R}), {b} batched solves", flush=True)
h = hs[0]; h.restore_state(); t = time.perf_counter(); h.optimize(10); print(f"solo {1e3 * (time.perf_counter() - t):.2f} ms")
