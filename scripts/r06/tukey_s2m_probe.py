"""round 6: the S2M-size Tukey start -- HIP runs (PCG + exact fallback at two tolerances, every solve exact) against the oracle"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from cuba_amd.capi import HipSolver
from cuba_amd.graph import flatten
from oracle.oracle import OracleSolver
from test_ref_lm import rejected_trial_cases
name = sys.argv[1] if len(sys.argv) > 1 else "s2m_lm10m_tukey"
make, rk, iters = rejected_trial_cases()[name]
g = make(); fp = flatten(g)
t = time.time(); o = OracleSolver(fp, rk); ro = o.optimize(iters); print(f"oracle {time.time() - t:.1f} s, trials {ro['trials'].tolist()}", flush=True)
for label, opts in (("tight", dict(pcg_tol=1e-11)), ("default", dict()), ("exact always", dict(reduced_solver=1))):
    h = HipSolver(fp, rk, **opts); t = time.time(); rh = h.optimize(iters)["chi2"]; dt = time.time() - t
    it, bad = h.pcg_history()
    dev = np.abs(rh / ro["chi2"] - 1) if len(rh) == len(ro["chi2"]) else np.array([np.inf])
    print(f"{label}: {dt:.3f} s, chi2 vs oracle max {dev.max():.2e} per iteration {' '.join(f'{v:.1e}' for v in dev)}; trials {h.counters()['lm_trials']}, exact {h.counter('exact_solve_fallbacks')}, pcg iterations {it.tolist()}", flush=True)
    for nm, a, b in zip("qtX", h.state(), o.state()):
        dd = np.abs(a - b).max(1)
        print(f"    {nm}: max abs diff vs oracle {dd.max():.2e}; percentiles 50 / 90 / 99 / 99.9: " + " / ".join(f"{np.percentile(dd, q):.1e}" for q in (50, 90, 99, 99.9))
              + f"; rows above 1e-3: {int((dd > 1e-3).sum())} of {len(dd)}", flush=True)
    h.close()
