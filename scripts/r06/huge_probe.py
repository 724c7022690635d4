"""A trajectory far beyond the BASELINE shapes (120 000 poses / 600 000 landmarks / 2.4 M edges): does the default path run, what does it cost, and does the exact reduced solver take it"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cuba_amd import capi
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_ba
P, L, E = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (120000, 600000, 2400000)
t0 = time.time(); fp = flatten(synth_ba(P, L, E, seed=12)); print(f"graph: {fp.Pt} poses, {fp.Lt} landmarks, {fp.E} edges ({time.time() - t0:.1f} s to generate)", flush=True)
rk = ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815))))
h = capi.HipSolver(fp, rk)
t0 = time.perf_counter(); r = h.optimize(5); t1 = time.perf_counter()
c = h.counters()
print(f"default path: 5 iterations in {1e3 * (t1 - t0):.1f} ms (first run on the structure), chi2 {r['chi2']}, pcg iterations {c['pcg_iterations']}, coarse dimension {c['coarse_dim']}, blocks {c['hsc_blocks']}, exact-solve fallbacks {h.counter('exact_solve_fallbacks')}, unconverged {h.pcg_history()[1]}", flush=True)
assert np.all(np.diff(r["chi2"]) < 0)
h.close()
e = capi.HipSolver(fp, rk, reduced_solver=1)
t0 = time.perf_counter(); re_ = e.optimize(3); t1 = time.perf_counter()
print(f"exact reduced solver: 3 iterations in {1e3 * (t1 - t0):.1f} ms (with the symbolic phase), chi2 {re_['chi2']}, exact solves {e.counter('exact_solve_fallbacks')}, failures {e.counter('exact_solve_failures')}", flush=True)
n = min(len(r["chi2"]), len(re_["chi2"]))
print("chi2 of the two paths, relative difference per iteration:", np.abs(r["chi2"][:n] - re_["chi2"][:n]) / re_["chi2"][:n])
