"""PCG tolerance sweep on the KITTI-00 shape: iterations, wall, chi2 parity and estimate RMSE vs the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from cuba_amd.capi import HipSolver
from oracle.oracle import OracleSolver
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
fp = flatten(synth_named("kitti00"))
o = OracleSolver(fp, RK); ref = o.optimize(10)["chi2"]; qo, to, Xo = o.state()
for tol in (1e-10, 1e-8, 1e-6, 1e-5, 1e-4):
    h = HipSolver(fp, RK, pcg_tol=tol); h.build_structure(); q0, t0, X0 = h.state()
    h.optimize(10); h.set_state(q0, t0, X0); c0 = h.counters()
    t = time.perf_counter(); got = h.optimize(10)["chi2"]; dt = time.perf_counter() - t
    q, tt, X = h.state()
    print("tol %.0e iters %5d wall %.1f ms chi2 max rel diff %.2e  rmse t %.2e X %.2e" % (tol, h.counters()["pcg_iterations"] - c0["pcg_iterations"], dt * 1e3,
          np.max(np.abs(got - ref) / ref), np.sqrt(((tt - to) ** 2).sum(1).mean()), np.sqrt(((X - Xo) ** 2).sum(1).mean())))
