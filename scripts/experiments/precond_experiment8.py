"""CPU experiment (oracle only): what would a MULTIPLICATIVE use of the same coarse space buy?  Same aggregates, same constant +
linear coarse functions as the built two-level preconditioner; compared are the iteration counts of
  additive      M^-1 = D^-1 + Q                       (built: Q = P (P^T A P)^-1 P^T, D = block-Jacobi)         1 SpMV / iteration
  A-DEF2        M^-1 = (I - Q A) D^-1 + Q , x0 = Q b  (deflation, Tang et al. 2009)                              2 SpMV
  hybrid / BNN  M^-1 = Q + (I - Q A) D^-1 (I - A Q)                                                              3 SpMV
  additive with a two-step Jacobi polynomial as the fine level, D^-1 (2 I - A D^-1) + Q                          2 SpMV
An iteration of the built solver is 12.3 us (5.9 SpMV + 6.4 fused update / coarse level); an extra SpMV-like pass costs ~6 us and an
extra coarse solve ~6 us more."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from oracle.oracle import OracleSolver
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
shape = sys.argv[1] if len(sys.argv) > 1 else "kitti00"
agg = int(sys.argv[2]) if len(sys.argv) > 2 else 24
its = [int(a) for a in sys.argv[3].split(',')] if len(sys.argv) > 3 else [3, 6, 9]
fp = flatten(synth_named(shape))
o = OracleSolver(fp, RK); o.build_structure()
q0, t0, X0 = o.state()
lams = o.optimize(10)["lambdas"]
P = fp.Pf; n = 6 * P

def system(i):
    o.set_state(q0, t0, X0)
    if i: o.optimize(i)
    o.compute_errors(); o.build_system(); o.set_lambda(lams[i]); o.schur()
    rp, ci, v = o.hsc()
    rows = np.repeat(np.arange(P), np.diff(rp))
    data = np.concatenate([v, v[rows != ci].transpose(0, 2, 1)])
    r_all = np.concatenate([rows, ci[rows != ci]]); c_all = np.concatenate([ci, rows[rows != ci]])
    order = np.lexsort((c_all, r_all))
    indptr = np.concatenate([[0], np.cumsum(np.bincount(r_all, minlength=P))])
    return sp.bsr_matrix((data[order], c_all[order], indptr), shape=(n, n)).tocsr(), o.array("bsc").copy().ravel()

def pcg(A, b, Minv, x0=None, tol=1e-7, maxit=3000):
    x = np.zeros_like(b) if x0 is None else x0.copy()
    r = b - A @ x; z = Minv(r); p = z.copy(); rz = r @ z
    rz0 = b @ Minv(b)                      # the built solver's reference: r0 = b
    k = 0
    while k < maxit and rz > tol * tol * rz0:
        q = A @ p; a = rz / (p @ q); x += a * p; r -= a * q; z = Minv(r); rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn; k += 1
    return k, x

idx = np.arange(P)
J = idx // agg
nc = int(J.max()) + 1
cnt = np.bincount(J, minlength=nc)
w = (2 * (idx % agg) + 1 - agg) / agg
rows, cols, vals = [], [], []
for c in range(6):
    rows += [6 * idx + c, 6 * idx + c]; cols += [12 * J + c, 12 * J + 6 + c]; vals += [np.ones(P), w]
Pm = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, 12 * nc))
for it in its:
    A, b = system(it)
    Dinv = np.linalg.inv(np.stack([A[6*j:6*j+6, 6*j:6*j+6].toarray() for j in range(P)]))
    jac = lambda r: np.einsum("nij,nj->ni", Dinv, r.reshape(P, 6)).ravel()
    Ac = (Pm.T @ A @ Pm).toarray()
    keep = np.abs(np.diag(Ac)) > 0                       # (a one-pose last aggregate has no linear function)
    Aci = np.zeros_like(Ac); Aci[np.ix_(keep, keep)] = np.linalg.inv(Ac[np.ix_(keep, keep)])
    Q = lambda r: Pm @ (Aci @ (Pm.T @ r))
    xs = np.linalg.solve(A.toarray(), b) if n <= 9000 else None
    res = {}
    res["additive"] = pcg(A, b, lambda r: jac(r) + Q(r))
    res["A-DEF2"] = pcg(A, b, lambda r: (lambda y: y - Q(A @ y))(jac(r)) + Q(r), x0=Q(b))
    def hybrid(r):
        y = Q(r); s = r - A @ y; z = jac(s); return y + z - Q(A @ z)
    res["hybrid"] = pcg(A, b, hybrid)
    res["additive, 2-step Jacobi"] = pcg(A, b, lambda r: (lambda y: 2 * y - jac(A @ y))(jac(r)) + Q(r))
    line = "  ".join("%s %d" % (k, v[0]) + ("" if xs is None else " (err %.1e)" % (np.abs(v[1] - xs).max() / np.abs(xs).max())) for k, v in res.items())
    print("%s LM iteration %d, lambda %.3g, %d-pose aggregates (Nc %d): %s" % (shape, it, lams[it], agg, 12 * nc, line), flush=True)
