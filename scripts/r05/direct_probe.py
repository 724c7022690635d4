"""round 5: the exact reduced solve (csrc/ba_direct.hip).  (1) time of one exact solve per shape (direct_after = 4 forces every solve of a
run there) and the chi2 it gives against the default PCG run; (2) the KITTI-00-size Tukey start whose PCG cannot finish."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from cuba_amd.capi import HipSolver
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_named
from test_ref_lm import rough_start
RK_TUKEY = ((2, 4.0), (2, 5.0))
RK = ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815))))
which = sys.argv[1]
if which == "tukey":
    fp = flatten(rough_start(synth_named("kitti00"), seed=1, sx=10.0, st=0.0, sr=0.0))
    for label, opts in (("default", dict()), ("tol 1e-11", dict(pcg_tol=1e-11)), ("direct_after 256", dict(direct_after=256)), ("direct_after 64", dict(direct_after=64)),
                        ("no fallback, max_iter 3000", dict(direct_fallback=0, pcg_max_iter=3000))):
        h = HipSolver(fp, RK_TUKEY, **opts)
        t = time.time(); r = h.optimize(14)["chi2"]; dt = time.time() - t
        it, bad = h.pcg_history()
        print(label, "unconverged", bad, "iters", it.tolist(), "trials", h.counters()["lm_trials"], "exact", h.counter("exact_solve_fallbacks"), h.counter("exact_solve_failures"),
              "%.3fs" % dt, "chi2", r.tolist(), flush=True)
        h.close()
else:
    fp = flatten(synth_named(which))
    h = HipSolver(fp, RK); t = time.time(); r0 = h.optimize(10)["chi2"]; dt0 = time.time() - t; h.close()
    h = HipSolver(fp, RK, direct_after=4); t = time.time(); r1 = h.optimize(10)["chi2"]; dt1 = time.time() - t
    print(which, "PCG run %.1f ms, exact-solve run %.1f ms (%d exact solves, %d failed), chi2 max rel diff %.2e" % (1e3 * dt0, 1e3 * dt1, h.counter("exact_solve_fallbacks"),
          h.counter("exact_solve_failures"), float(np.abs(r1 / r0 - 1).max()) if len(r1) == len(r0) else np.inf), flush=True)
    t = time.time(); r2 = h.optimize(10)["chi2"]; print(which, "second exact-solve run %.1f ms" % (1e3 * (time.time() - t)))
    h.close()
