import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from cuba_amd import capi
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_ba
from oracle.oracle import OracleSolver
rng = np.random.default_rng(11); bad = 0; n = 0
rk = ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815))))
for it in range(40):
    P = int(rng.choice([4, 9, 30, 77, 200, 420, 900])); L = int(rng.integers(8 * P, 30 * P + 40)); E = int(L * rng.uniform(2.8, 5.0))
    try: g = synth_ba(P, L, E, seed=int(rng.integers(1 << 30)))
    except (RuntimeError, ValueError): continue
    fp = flatten(g)
    key = fp.eP.astype(np.int64) * (fp.Lt + 1) + fp.eL
    if len(np.unique(key)) != fp.E or fp.Pt == fp.Pf: continue
    o = OracleSolver(fp, rk); ro = o.optimize(6)["chi2"]
    for mode, opts in (("f32 exact", dict(reduced_solver=1)), ("f32 pcg", {}), ("f64 mixed exact", dict(reduced_solver=1, mixed_precision=1))):
        h = capi.HipSolver(fp, rk, precision="f32" if mode.startswith("f32") else "f64", **opts); r = h.optimize(6)["chi2"]
        m = min(len(r), len(ro)); d = float(np.max(np.abs(r[:m] - ro[:m]) / ro[:m])) if m else 1.0
        n += 1
        if m < 4 or d > (1e-3 if mode.startswith("f32") else 1e-6):
            bad += 1; print("FAIL", mode, f"P {fp.Pt} L {fp.Lt} E {fp.E}: lengths {len(r)} {len(ro)} chi2 rel {d:.2e}", h.counters().get("exact_solve_fallbacks"), flush=True)
        h.close()
print(f"{n} runs, {bad} failures")
