"""round 6: the sparse exact reduced solve (csrc/ba_direct.hip).  (1) kernels against LAPACK on dense and sparse patterns, (2) per shape:
one exact solve against a PCG solve at pcg_tol 1e-12 (increment), time per solve, LM run with every solve exact against the PCG run."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from cuba_amd.capi import HipSolver, dense_solve
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_named, synth_ba
from sparse_chol_emulator import random_spd_blocks
from test_sparse_plan import band_pattern

RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
what = sys.argv[1:] or ["kernels", "kitti07", "kitti00"]

if "kernels" in what:
    for n in (6, 30, 126, 132, 384, 1482):
        rng = np.random.default_rng(n)
        Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
        A = (Q * np.logspace(0, 8, n)) @ Q.T; A = 0.5 * (A + A.T)
        b = rng.normal(size=n)
        x, bad, st = dense_solve(A, b, with_stats=True)
        ref = np.linalg.solve(A, b)
        print(f"dense n={n}: bad={bad} resid {np.abs(A @ x - b).max() / (np.abs(A).max() * np.abs(x).max() + np.abs(b).max()):.2e} err {np.abs(x - ref).max() / np.abs(ref).max():.2e} {st}", flush=True)
    for name, (rp, ci) in {"band_120": band_pattern(120, 9), "loop": band_pattern(203, 7, closures=[(0, 150, 40)]), "two": band_pattern(300, 5, closures=[(10, 200, 12), (60, 280, 15)]),
                           "traj1000": band_pattern(1000, 18, closures=[(0, 770, 230)])}.items():
        rng = np.random.default_rng(len(ci))
        A = random_spd_blocks(rp, ci, rng); b = rng.normal(size=A.shape[0])
        ref = np.linalg.solve(A, b)
        for slack in (-1, 0, 4):
            x, bad, st = dense_solve(A, b, slack=slack, with_stats=True)
            x2, _ = dense_solve(A, b, slack=slack)
            print(f"sparse {name} slack {slack}: bad={bad} err {np.abs(x - ref).max() / np.abs(ref).max():.2e} deterministic {np.array_equal(x, x2)} {st}", flush=True)
        x32, bad32 = dense_solve(A, b, precision="f32")
        print(f"sparse {name} f32: bad={bad32} err {np.abs(x32 - ref).max() / np.abs(ref).max():.2e}", flush=True)

for shape in [w for w in what if w != "kernels"]:
    fp = flatten(synth_named(shape))
    a = HipSolver(fp, RK, pcg_tol=1e-12, direct_fallback=0); md = a.max_diagonal(); lam = 1e-5 * md
    a.set_lambda(lam); assert a.solve(); xa = a.array("xp")
    b = HipSolver(fp, RK, reduced_solver=1); b.max_diagonal(); b.set_lambda(lam)
    assert b.solve(); xb = b.array("xp")
    print(f"{shape}: exact vs PCG(1e-12) increment: max rel diff {np.abs(xa - xb).max() / np.abs(xa).max():.2e}; exact solves {b.counter('exact_solve_fallbacks')}", flush=True)
    ts = []
    for _ in range(5):
        t = time.perf_counter(); assert b.solve(); b.array("xp")[:1]; ts.append(time.perf_counter() - t)
    print(f"{shape}: schur + exact solve + back-substitution wall {1e3 * min(ts):.3f} ms (min of 5)", flush=True)
    for label, opts in (("pcg", dict()), ("exact", dict(reduced_solver=1))):
        h = HipSolver(fp, RK, **opts); h.optimize(10)
        q0 = fp.q, fp.t, fp.Xw
        best = 1e9
        for _ in range(3):
            h.set_state(*q0); t = time.perf_counter(); r = h.optimize(10)["chi2"]; best = min(best, time.perf_counter() - t)
        print(f"{shape} [{label}]: optimize(10) {1e3 * best:.2f} ms, chi2[-1] {r[-1]:.9e}, trials {h.counters()['lm_trials']}, exact solves {h.counter('exact_solve_fallbacks')}", flush=True)
        if label == "pcg": rp = r
        else: print(f"{shape}: exact-run chi2 vs pcg-run chi2 max rel diff {np.abs(r / rp - 1).max():.2e}", flush=True)
        h.close()
