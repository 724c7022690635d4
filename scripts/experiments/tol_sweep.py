"""PCG tolerance x coarse-inverse age sweep: iterations, wall, chi2 parity and estimate RMSE vs the oracle.
   python scripts/experiments/tol_sweep.py [shape]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from cuba_amd.capi import HipSolver
from oracle.oracle import OracleSolver
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
shape = sys.argv[1] if len(sys.argv) > 1 else "kitti00"
fp = flatten(synth_named(shape))
o = OracleSolver(fp, RK); ref = o.optimize(10)["chi2"]; qo, to, Xo = o.state()
for tol, age in ((1e-10, 2), (1e-8, 2), (1e-7, 2), (1e-6, 2), (1e-5, 2), (1e-8, 0), (1e-8, 3), (1e-8, 4), (1e-8, 9), (1e-7, 4), (1e-7, 9), (1e-6, 4), (1e-6, 9)):
    h = HipSolver(fp, RK, pcg_tol=tol, coarse_max_age=age); h.build_structure(); q0, t0, X0 = h.state()
    h.optimize(10); h.set_state(q0, t0, X0); c0 = h.counters()
    t = time.perf_counter(); got = h.optimize(10)["chi2"]; dt = time.perf_counter() - t
    q, tt, X = h.state()
    print("tol %.0e age %d iters %5d wall %.2f ms chi2 max rel diff %.2e  rmse t %.2e X %.2e" % (tol, age, h.counters()["pcg_iterations"] - c0["pcg_iterations"], dt * 1e3,
          np.max(np.abs(got - ref) / ref), np.sqrt(((tt - to) ** 2).sum(1).mean()), np.sqrt(((X - Xo) ** 2).sum(1).mean())), flush=True)
    h.close()
