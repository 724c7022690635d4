"""Stage-by-stage comparison HIP vs oracle on one draw of parity_fuzz.py: python draw_stage_probe.py <draw>"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cuba_amd import capi
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_ba
from oracle.oracle import OracleSolver
def draw(target):
    rng = np.random.default_rng(2026)
    KINDS = [((0, 0.0), (0, 0.0)), ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815)))), ((2, 4.0), (2, 5.0)), ((1, 2.0), (2, 6.0))]
    for it in range(1000):
        P = int(rng.choice([3, 5, 8, 13, 24, 40, 77, 130, 260, 420]))
        L = int(rng.integers(max(20, 4 * P), 40 * P + 50)); E = int(L * rng.uniform(2.2, 6.0))
        try:
            g = synth_ba(P, L, E, seed=int(rng.integers(1 << 30)), stereo_frac=float(rng.choice([0.0, 0.3, 0.85, 1.0])), outlier_frac=float(rng.choice([0.0, 0.03, 0.1])), loop_closure=bool(rng.integers(2)))
        except (RuntimeError, ValueError):
            continue
        g = copy.deepcopy(g)
        if rng.random() < 0.5:
            k = int(rng.integers(1, max(2, P // 4)))
            g.pose_fixed[rng.choice(P, size=min(k, P - 1), replace=False)] = True
        if rng.random() < 0.4:
            g.lm_fixed[rng.choice(g.nlandmarks, size=g.nlandmarks // int(rng.integers(3, 20)), replace=False)] = True
        try: fp = flatten(g)
        except Exception: continue
        if fp.E == 0 or (fp.Pf == 0 and fp.Lf == 0): continue
        rk = KINDS[int(rng.integers(len(KINDS)))]
        iters = int(rng.integers(3, 9))
        if it == target: return g, fp, rk, iters
def rel(a, b): return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))
def sym6(u):
    m = np.zeros((len(u), 3, 3)); idx = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    for k, (i, j) in enumerate(idx): m[:, i, j] = u[:, k]; m[:, j, i] = u[:, k]
    return m
g, fp, rk, iters = draw(int(sys.argv[1]))
o, h = OracleSolver(fp, rk), capi.HipSolver(fp, rk, pcg_tol=1e-12)
print("errors", h.compute_errors(), o.compute_errors())
o.build_system(); h.build_system()
md = o.max_diagonal(); print("max diagonal", h.max_diagonal(), md)
lm = h.array("lm_sys").reshape(-1, 9)
print("Hll", rel(sym6(lm[:, :6]), o.array("Hll").reshape(-1, 3, 3)), "bl", rel(lm[:, 6:], o.array("bl").reshape(-1, 3)), "bp", rel(h.array("bp"), o.array("bp")))
lam = 1e-5 * md
o.set_lambda(lam); h.set_lambda(lam); o.schur(); h.schur()
rpo, cio, vo = o.hsc(); rp, ci, v = h.hsc()
print("pattern equal", np.array_equal(rp, rpo) and np.array_equal(ci, cio), "nblk", len(ci), len(cio))
if np.array_equal(rp, rpo) and np.array_equal(ci, cio):
    diag = np.zeros(len(ci), bool); diag[rp[:-1]] = True; iu = np.triu_indices(6)
    print("Hsc off-diagonal", rel(v[~diag], vo[~diag]) if (~diag).any() else 0, "diagonal", rel(v[diag][:, iu[0], iu[1]] + lam * (iu[0] == iu[1]), vo[diag][:, iu[0], iu[1]]))
    d = np.abs(v[~diag] - vo[~diag]).reshape(-1, 36).max(1); w = np.argsort(d)[-5:]
    rows = np.repeat(np.arange(fp.Pf), np.diff(rp))[~diag]; cols = ci[~diag]
    print("worst off-diagonal blocks (row, col, abs diff, oracle magnitude):", [(int(rows[k]), int(cols[k]), float(d[k]), float(np.abs(vo[~diag][k]).max())) for k in w])
print("bsc", rel(h.array("bsc"), o.array("bsc")))
print("solve", o.solve(), h.solve_reduced()); h.back_substitute()
print("xp", rel(h.array("xp"), o.array("xp")), "xl", rel(h.array("xl"), o.array("xl")))
# duplicate observations?
key = fp.eP.astype(np.int64) * (fp.Lt + 1) + fp.eL
print("duplicate (pose, landmark) observations:", len(key) - len(np.unique(key)))
u, c = np.unique(key, return_counts=True)
for k in u[c > 1]:
    p_, l_ = int(k // (fp.Lt + 1)), int(k % (fp.Lt + 1))
    ids = np.nonzero(key == k)[0]
    print("duplicate: pose", p_, "(free)" if p_ < fp.Pf else "(fixed)", "landmark", l_, "(free)" if l_ < fp.Lf else "(fixed)", "edges", ids.tolist(), "count", len(ids), "dims", fp.eDim[ids].tolist())
if np.array_equal(rp, rpo):
    dd = np.abs((v[diag][:, iu[0], iu[1]] + lam * (iu[0] == iu[1])) - vo[diag][:, iu[0], iu[1]]).max(1) / np.abs(vo[diag]).reshape(-1, 36).max(1)
    print("diagonal blocks off by more than 1e-9:", [(int(i), float(dd[i])) for i in np.nonzero(dd > 1e-9)[0]])
for opts in (dict(device_setup=0), dict(landmark_reorder=0), dict(pose_reorder=0)):
    h2 = capi.HipSolver(fp, rk, pcg_tol=1e-12, **opts); h2.build_system(); h2.max_diagonal(); h2.set_lambda(lam); h2.schur()
    rp2, ci2, v2 = h2.hsc(); d2 = np.zeros(len(ci2), bool); d2[rp2[:-1]] = True
    print(opts, "Hsc diagonal", rel(v2[d2][:, iu[0], iu[1]] + lam * (iu[0] == iu[1]), vo[diag][:, iu[0], iu[1]]))
