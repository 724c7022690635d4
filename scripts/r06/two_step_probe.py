import sys, copy
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from cuba_amd.capi import HipSolver
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_ba
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
fp = flatten(synth_ba(40, 600, 2400, seed=1))
for rep in range(4):
    for opts in ({}, dict(device_setup=0)):
        a = HipSolver(fp, RK, **opts); ra = a.optimize(4)["chi2"]
        b = HipSolver(None, RK, **opts); b.set_graph(fp, two_step=True); rb = b.optimize(4)["chi2"]
        fp2 = copy.copy(fp); fp2.meas = fp.meas + 0.25
        a.set_graph(fp2); b.set_graph(fp2, two_step=True)
        xa = a.optimize(3)["chi2"]; xb = b.optimize(3)["chi2"]
        print(rep, opts, np.array_equal(ra, rb), np.array_equal(xa, xb), xa.tolist(), xb.tolist(), a.pcg_history()[0].tolist(), b.pcg_history()[0].tolist(), a.counters()["lm_trials"], b.counters()["lm_trials"], flush=True)
        a.close(); b.close()
