"""(needs the option pcg_forcing, which existed in commit c79419a only)  round 4 experiment: forcing term of the inexact LM step (option pcg_forcing) against the stated estimate tolerances.
For every BASELINE shape: the oracle once (exact solves), then the HIP path for (pcg_tol, pcg_forcing) pairs: chi2 max-rel-diff per
iteration, RMSE of the final estimates (as tests/test_gpu_configs.py takes it), PCG iterations per run, wall of a 10-iteration run."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cuba_amd.capi import HipSolver
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_named
from oracle.oracle import OracleSolver
RK = ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815))))
def rmse(a, b):
    d = np.asarray(a) - np.asarray(b); return float(np.sqrt((d * d).sum(1).mean()))
for name in sys.argv[1:]:
    fp = flatten(synth_named(name))
    o = OracleSolver(fp, RK, threads=1); ref = o.optimize(10)["chi2"]; rs = o.state()
    for tol, eta in ((1e-7, 0), (1e-7, 1), (1e-7, 0.3), (1e-7, 0.1), (3e-8, 1), (1e-8, 1), (1e-8, 3), (1e-8, 10)):
        h = HipSolver(fp, RK, pcg_tol=tol, pcg_forcing=eta)
        got = h.optimize(10)["chi2"]
        st = h.state()
        it = h.pcg_history()[0]
        h.set_state(fp.q, fp.t, fp.Xw); h.optimize(10)           # second run: graphs exist
        h.set_state(fp.q, fp.t, fp.Xw)
        t = time.perf_counter(); h.optimize(10); dt = time.perf_counter() - t
        print(f"{name} tol {tol:g} eta {eta:g}: chi2 {np.abs(got / ref - 1).max():.2e}  q {rmse(st[0], rs[0]):.2e} t {rmse(st[1], rs[1]):.2e} X {rmse(st[2], rs[2]):.2e}  "
              f"its {int(it.sum())} {it.tolist()}  wall {dt * 1e3:.2f} ms", flush=True)
        h.close()
