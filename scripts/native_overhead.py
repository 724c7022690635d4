"""Overhead of the native multi-GPU driver at N = 1: cuba_hip_dist_optimize over a 1-rank RCCL communicator against cuba_hip_optimize
on the same handle configuration (what the landmark-partitioned mode costs before any communication happens)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from cuba_amd.capi import HipSolver
from cuba_amd.dist import NativeDist, rccl_unique_id
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
for shape in sys.argv[1:] or ["kitti00"]:
    fp = flatten(synth_named(shape))
    h = HipSolver(fp, RK); h.build_structure(); q0, t0, X0 = h.state(); h.optimize(10)
    ta = []
    for _ in range(8):
        h.set_state(q0, t0, X0); t = time.perf_counter(); a = h.optimize(10)["chi2"]; ta.append(time.perf_counter() - t)
    h2 = HipSolver(fp, RK); d = NativeDist(h2, fp, 0, 1, unique_id=rccl_unique_id()); d.optimize(10)
    tb = []
    for _ in range(8):
        h2.set_state(q0, t0, X0); t = time.perf_counter(); b = d.optimize(10); tb.append(time.perf_counter() - t)
    print(f"{shape}: cuba_hip_optimize {min(ta)*1e3:.2f} ms, cuba_hip_dist_optimize (world = 1, RCCL communicator, host structure pipeline) "
          f"{min(tb)*1e3:.2f} ms, chi2 identical: {bool(np.array_equal(a, b))}, counters {d.counters()}, PCG iterations of the last run {int(np.abs(h.pcg_history()[0][-10:]).sum())} vs {int(np.abs(h2.pcg_history()[0][-10:]).sum())}", flush=True)
