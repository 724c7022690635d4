"""round 4 diagnostics: (1) the K00 Tukey / 10 m-landmark-noise case whose tight-tolerance run hit pcg_max_iter; (2) r0.z0 per solve."""
import os, sys, copy, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from cuba_amd.capi import HipSolver
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_named
from test_ref_lm import rough_start
RK_TUKEY = ((2, 4.0), (2, 5.0))
which = sys.argv[1]
if which == "tukey":
    fp = flatten(rough_start(synth_named("kitti00"), seed=1, sx=10.0, st=0.0, sr=0.0))
    for label, opts in (("tol 1e-11", dict(pcg_tol=1e-11)), ("tol 1e-11 fp64 inverse", dict(pcg_tol=1e-11, precond_fp32=0)),
                        ("tol 1e-10", dict(pcg_tol=1e-10)), ("tol 1e-9", dict(pcg_tol=1e-9)), ("default", dict())):
        h = HipSolver(fp, RK_TUKEY, pcg_max_iter=3000, **opts)
        t = time.time(); r = h.optimize(14)["chi2"]; dt = time.time() - t
        it, bad = h.pcg_history()
        print(label, "unconverged", bad, "iters", it.tolist(), "trials", h.counters()["lm_trials"], "%.2fs" % dt, "chi2", r[-1] if len(r) else None, flush=True)
        h.close()
else:
    RK = ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815))))
    fp = flatten(synth_named(which))
    h = HipSolver(fp, RK)
    h.optimize(10)
    print(which, h.pcg_history()[0].tolist())
