"""round 5: the exact reduced solve at S2M / G4M size (dense: 7 / 29 GB) -- one forced solve each, against the PCG's increment"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cuba_amd.capi import HipSolver
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_named
RK = ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815))))
for shape in sys.argv[1:]:
    fp = flatten(synth_named(shape))
    a = HipSolver(fp, RK, pcg_tol=1e-12); lam = 1e-5 * a.max_diagonal(); a.set_lambda(lam); assert a.solve(); xa = a.array("xp"); a.close()
    b = HipSolver(fp, RK, direct_after=4); b.max_diagonal(); b.set_lambda(lam)
    t = time.perf_counter(); ok = b.solve(); dt = time.perf_counter() - t
    xb = b.array("xp")
    print(f"{shape}: exact solve ok={ok} ({b.counter('exact_solve_fallbacks')} fallbacks, {b.counter('exact_solve_failures')} failures), wall of schur + 4 PCG iterations + exact solve {dt:.3f} s, "
          f"max |dx - dx_pcg(1e-12)| / max |dx| = {np.abs(xa - xb).max() / np.abs(xa).max():.2e}", flush=True)
    t = time.perf_counter(); ok = b.solve(); dt = time.perf_counter() - t
    print(f"{shape}: second exact solve (matrix allocated) {dt:.3f} s", flush=True)
    b.close()
