"""Randomised parity sweep, HIP path against the oracle (not part of the suite: ~60 graphs of random size, edge mix, fixed vertices, robust
kernels; default PCG path at the stated bars + the exact reduced solver at 1e-9; fp64 library; batched execution of mixed graphs against solo)."""
import os, sys, copy, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cuba_amd import capi
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_ba
from oracle.oracle import OracleSolver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
only = int(sys.argv[2]) if len(sys.argv) > 2 else -1       # details of one draw
sizes = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [3, 5, 8, 13, 24, 40, 77, 130, 260, 420]      # pose counts to draw from
rng = np.random.default_rng(2026)
KINDS = [((0, 0.0), (0, 0.0)), ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815)))), ((2, 4.0), (2, 5.0)), ((1, 2.0), (2, 6.0))]
bad = 0; soft = 0; skipped = 0; used = 0; worst = dict(chi=0.0, q=0.0, t=0.0, X=0.0, chi_exact=0.0)
kept = []
for it in range(n):
    P = int(rng.choice(sizes))
    L = int(rng.integers(max(20, 4 * P), 40 * P + 50))
    E = int(L * rng.uniform(2.2, 6.0))
    try:
        g = synth_ba(P, L, E, seed=int(rng.integers(1 << 30)), stereo_frac=float(rng.choice([0.0, 0.3, 0.85, 1.0])), outlier_frac=float(rng.choice([0.0, 0.03, 0.1])),
                     loop_closure=bool(rng.integers(2)))
    except (RuntimeError, ValueError):      # (the generator cannot place that many tracks on so few poses)
        continue
    g = copy.deepcopy(g)
    if rng.random() < 0.5:
        k = int(rng.integers(1, max(2, P // 4)))
        g.pose_fixed[rng.choice(P, size=min(k, P - 1), replace=False)] = True
    if rng.random() < 0.4:
        g.lm_fixed[rng.choice(g.nlandmarks, size=g.nlandmarks // int(rng.integers(3, 20)), replace=False)] = True
    try:
        fp = flatten(g)
    except Exception as e:       # (a draw that leaves no free vertex / no edge)
        continue
    if fp.E == 0 or (fp.Pf == 0 and fp.Lf == 0):
        continue
    # two classes of draws are left out: graphs with a landmark observed twice by one pose (the generator's loop-closure window makes a few) --
    # there the library forms the complete Schur complement and the reference / the oracle add only one of the two cross terms (DESIGN section 5,
    # test_duplicate_observations_of_one_pose checks the library against a dense solve instead) -- and graphs whose gauge is free (no fixed pose, or
    # monocular only with a single fixed pose and no fixed landmark: the scale), where the trajectories of two exact solvers part by rounding
    key = fp.eP.astype(np.int64) * (fp.Lt + 1) + fp.eL
    gauge_fixed = fp.Pt > fp.Pf and (fp.E3 > 0 or fp.Lt > fp.Lf or fp.Pt - fp.Pf >= 2)
    if len(np.unique(key)) != fp.E or not gauge_fixed:
        skipped += 1
        continue
    used += 1
    rk = KINDS[int(rng.integers(len(KINDS)))]
    iters = int(rng.integers(3, 9))
    if only >= 0 and it != only: continue
    o = OracleSolver(fp, rk); ro = o.optimize(iters)
    label = f"#{it} P {fp.Pt} (free {fp.Pf}) L {fp.Lt} (free {fp.Lf}) E {fp.E} (stereo {fp.E3}) kernels {rk[0][0]}/{rk[1][0]} iters {iters}"
    for mode, opts, bars in (("pcg", {}, dict(chi=1e-6, q=1e-8, t=1e-6, X=1e-6)), ("exact", dict(reduced_solver=1), dict(chi=1e-9, q=1e-10, t=1e-8, X=1e-8))):
        h = capi.HipSolver(fp, rk, **opts); rh = h.optimize(iters)
        ok = len(rh["chi2"]) == len(ro["chi2"])
        chi = float(np.max(np.abs(rh["chi2"] - ro["chi2"]) / np.maximum(ro["chi2"], 1e-300))) if ok and len(ro["chi2"]) else 0.0
        est = [float(np.sqrt(((a - b) ** 2).sum(1).mean())) if len(a) else 0.0 for a, b in zip(h.state(), o.state())]
        fail = (not ok) or chi > bars["chi"] or est[0] > bars["q"] or est[1] > bars["t"] or est[2] > bars["X"] or h.pcg_history()[1] != 0
        if mode == "pcg":
            worst["chi"] = max(worst["chi"], chi); worst["q"] = max(worst["q"], est[0]); worst["t"] = max(worst["t"], est[1]); worst["X"] = max(worst["X"], est[2])
        else:
            worst["chi_exact"] = max(worst["chi_exact"], chi)
        if only >= 0:
            print(mode, "hip   ", rh["chi2"], rh.get("trials"), "\n", mode, "oracle", ro["chi2"], ro.get("trials"), ro.get("lambdas"))
            h2 = capi.HipSolver(fp, rk, pcg_tol=1e-12, **opts); r2 = h2.optimize(iters); print(mode, "hip pcg_tol 1e-12", r2["chi2"]); h2.close()
        if fail and mode == "pcg":
            # a miss at the default tolerance must be the tolerance's (small, weakly determined graphs amplify the PCG's 1e-7): the same run at
            # pcg_tol = 1e-10 has to meet the bars with room
            ht = capi.HipSolver(fp, rk, pcg_tol=1e-10); rt = ht.optimize(iters)
            chit = float(np.max(np.abs(rt["chi2"] - ro["chi2"]) / ro["chi2"])) if len(rt["chi2"]) == len(ro["chi2"]) else 1.0
            estt = [float(np.sqrt(((a - b) ** 2).sum(1).mean())) if len(a) else 0.0 for a, b in zip(ht.state(), o.state())]
            ht.close()
            tight_ok = chit <= 1e-9 and estt[0] <= 1e-10 and estt[1] <= 1e-8 and estt[2] <= 1e-8
            print(f"   at pcg_tol 1e-10: chi2 {chit:.2e} q {estt[0]:.2e} t {estt[1]:.2e} X {estt[2]:.2e} -> {'tolerance effect' if tight_ok else 'NOT explained by the tolerance'}")
            if tight_ok:
                soft += 1; fail = False
                print("soft", label, f"default tolerance: chi2 {chi:.2e} q {est[0]:.2e} t {est[1]:.2e} X {est[2]:.2e} handed over {h.pcg_history()[1]}", flush=True)
        if fail:
            bad += 1
            if not ok or chi > 100 * bars["chi"]: print("   hip   ", rh["chi2"], "\n   oracle", ro["chi2"], "\n   counters", {k: v for k, v in h.counters().items() if v})
            print("FAIL", mode, label, "trajectory lengths", len(rh["chi2"]), len(ro["chi2"]), f"chi2 {chi:.2e} q {est[0]:.2e} t {est[1]:.2e} X {est[2]:.2e} unconverged {h.pcg_history()[1]}", flush=True)
        h.close()
    if len(kept) < 12 and fp.Pf > 0 and fp.Lf > 0: kept.append((fp, rk, iters))
print(f"{n} draws, {used} used ({skipped} left out: duplicate observations or free gauge): {bad} failures, {soft} runs outside the default-tolerance bars that meet the tight bars at pcg_tol 1e-10; worst over the PCG runs: chi2 {worst['chi']:.2e} q {worst['q']:.2e} t {worst['t']:.2e} X {worst['X']:.2e}; exact-solver runs: chi2 {worst['chi_exact']:.2e}", flush=True)
# batched execution of the kept graphs (mixed sizes, kernels and iteration counts share one batch of common length) against solo runs
if kept:
    iters = 5
    solo = []
    for fp, rk, _ in kept:
        h = capi.HipSolver(fp, rk); solo.append((h.optimize(iters)["chi2"], h.state())); h.close()
    hs = [capi.HipSolver(fp, rk) for fp, rk, _ in kept]
    res, batched = capi.optimize_batch(hs, iters)
    same = all(np.array_equal(r, s[0]) and all(np.array_equal(a, b) for a, b in zip(h.state(), s[1])) for r, s, h in zip(res, solo, hs))
    print(f"batch of {len(kept)} mixed graphs bit-identical to solo: {same} ({batched} reduced solves ran batched)", flush=True)
