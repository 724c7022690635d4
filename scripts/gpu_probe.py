"""Exploratory GPU run: stage-by-stage HIP vs oracle on a small graph, then LM on KITTI shapes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cuba_amd.synth import synth_ba, synth_named
from cuba_amd.graph import flatten
from cuba_amd.capi import HipSolver
from oracle.oracle import OracleSolver

RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))

def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))

g = synth_ba(40, 600, 2400, seed=1)
fp = flatten(g)
o = OracleSolver(fp, RK); h = HipSolver(fp, RK)
print("chi2", o.compute_errors(), h.compute_errors())
o.build_system()
md_o, md_h = o.max_diagonal(), h.max_diagonal()
print("maxdiag", md_o, md_h)
lm = h.array("lm_sys").reshape(-1, 9)
Hll_o = o.array("Hll").reshape(-1, 3, 3)
print("Hll rel", rel(lm[:, [0, 1, 2, 3, 4, 5]], np.stack([Hll_o[:,0,0],Hll_o[:,0,1],Hll_o[:,0,2],Hll_o[:,1,1],Hll_o[:,1,2],Hll_o[:,2,2]],1)), "bl rel", rel(lm[:, 6:], o.array("bl").reshape(-1,3)))
print("bp rel", rel(h.array("bp"), o.array("bp")))
lam = 1e-5 * md_o
o.set_lambda(lam); h.set_lambda(lam)
o.schur(); h.schur()
rp_o, ci_o, v_o = o.hsc(); rp_h, ci_h, v_h = h.hsc()
print("pattern equal", np.array_equal(rp_o, rp_h), np.array_equal(ci_o, ci_h))
# diag blocks of HIP are upper-only before pcg setup: compare upper triangles for diagonal blocks, full for others
diag = np.zeros(len(ci_h), bool); diag[rp_h[:-1]] = True
iu = np.triu_indices(6)
print("Hsc offdiag rel", rel(v_h[~diag], v_o[~diag]), "diag(upper) rel", rel(v_h[diag][:, iu[0], iu[1]], v_o[diag][:, iu[0], iu[1]]))
print("bsc rel", rel(h.array("bsc"), o.array("bsc")))
ok_o = o.solve(); ok_h = h.solve_reduced(); h.back_substitute()
print("ok", ok_o, ok_h, "xp rel", rel(h.array("xp"), o.array("xp")), "xl rel", rel(h.array("xl"), o.array("xl")), h.counters())
print("scale", o.compute_scale(lam), h.compute_scale(lam))
o.update(); h.update()
qo, to, Xo = o.state(); qh, th, Xh = h.state()
print("state rel", rel(qh, qo), rel(th, to), rel(Xh, Xo))
print("chi2 after", o.compute_errors(), h.compute_errors())

for name in sys.argv[1:] or ["kitti07", "kitti00"]:
    g = synth_named(name); fp = flatten(g)
    o = OracleSolver(fp, RK)
    t0 = time.time(); ro = o.optimize(10); t_o = time.time() - t0
    h = HipSolver(fp, RK)
    h.build_structure()
    t0 = time.time(); rh = h.optimize(10); t_h = time.time() - t0
    n = min(len(ro["chi2"]), len(rh["chi2"]))
    print(name, "oracle %.3fs hip %.3fs" % (t_o, t_h), "iters", len(ro["chi2"]), len(rh["chi2"]))
    print(" chi2 rel per iter", np.abs(rh["chi2"][:n] - ro["chi2"][:n]) / ro["chi2"][:n])
    print(" counters", h.counters())
    qo, to, Xo = o.state(); qh, th, Xh = h.state()
    print(" rmse q %.3e t %.3e X %.3e" % (np.sqrt(((qo-qh)**2).sum(1).mean()), np.sqrt(((to-th)**2).sum(1).mean()), np.sqrt(((Xo-Xh)**2).sum(1).mean())))
    # timed again with profile buckets
    h2 = HipSolver(fp, RK, profile=1); h2.build_structure(); h2.optimize(10)
    print(" profile", {k: round(v*1e3, 2) for k, v in h2.profile().items()})
    # warm second run for timing without first-launch overheads
    h3 = HipSolver(fp, RK); h3.build_structure()
    t0 = time.time(); h3.optimize(10); print(" warm optimize(10) %.1f ms" % ((time.time()-t0)*1e3), h3.counters())
