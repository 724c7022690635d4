"""CPU experiment 2: coarse-space variants for the two-level PCG preconditioner on the KITTI-00-shaped system."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from oracle.oracle import OracleSolver

RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
fp = flatten(synth_named("kitti00"))
o = OracleSolver(fp, RK)
P = fp.Pf; n = 6 * P

def system(nit):
    res = o.optimize(nit) if nit else dict(lambdas=[None])
    o.compute_errors(); o.build_system()
    lam = res["lambdas"][-1] if nit else 1e-5 * o.max_diagonal()
    o.set_lambda(lam); o.schur()
    rp, ci, v = o.hsc(); b = o.array("bsc")
    rows = np.repeat(np.arange(P), np.diff(rp))
    data = np.concatenate([v, v[rows != ci].transpose(0, 2, 1)])
    r_all = np.concatenate([rows, ci[rows != ci]]); c_all = np.concatenate([ci, rows[rows != ci]])
    order = np.lexsort((c_all, r_all))
    indptr = np.concatenate([[0], np.cumsum(np.bincount(r_all, minlength=P))])
    return sp.bsr_matrix((data[order], c_all[order], indptr), shape=(n, n)).tocsr(), b, lam

def pcg(A, b, Minv, tol=1e-10, maxit=5000):
    x = np.zeros_like(b); r = b.copy(); z = Minv(r); p = z.copy(); rz = r @ z; rz0 = rz; k = 0
    while k < maxit and rz > tol * tol * rz0:
        q = A @ p; a = rz / (p @ q); x += a * p; r -= a * q; z = Minv(r); rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn; k += 1
    return x, k

def jac_of(A):
    Dinv = np.linalg.inv(np.stack([A[6*i:6*i+6, 6*i:6*i+6].toarray() for i in range(P)]))
    return lambda r: np.einsum("nij,nj->ni", Dinv, r.reshape(P, 6)).ravel()

def prolong(g, linear):
    nb = (P + g - 1) // g; agg = np.arange(P) // g
    rows = np.arange(n); comp = np.tile(np.arange(6), P); a6 = np.repeat(agg, 6)
    if not linear:
        return sp.csr_matrix((np.ones(n), (rows, 6 * a6 + comp)), shape=(n, 6 * nb))
    pos = np.repeat((np.arange(P) - (agg * g + (g - 1) / 2)) / (g / 2), 6)
    return sp.csr_matrix((np.concatenate([np.ones(n), pos]), (np.concatenate([rows, rows]), np.concatenate([12 * a6 + comp, 12 * a6 + 6 + comp]))), shape=(n, 12 * nb))

Aprev = None
for nit in (0, 3, 6, 9):
    A, b, lam = system(nit if nit == 0 else 3)   # optimize() continues from the current state: 3 more iterations each time
    jac = jac_of(A)
    out = [f"LM it {nit} lam {lam:.3g}"]
    x, k = pcg(A, b, jac); out.append(f"jacobi {k}")
    for g, lin in ((16, False), (32, True), (24, True), (16, True)):
        Pm = prolong(g, lin); Aci = np.linalg.inv((Pm.T @ A @ Pm).toarray())
        x, k = pcg(A, b, lambda r: jac(r) + Pm @ (Aci @ (Pm.T @ r))); out.append(f"g{g}{'lin' if lin else 'const'}(dim {Pm.shape[1]}) {k}")
        if g == 16 and not lin:
            if Aprev is not None:
                x, k2 = pcg(A, b, lambda r: jac(r) + Pm @ (Aprev @ (Pm.T @ r))); out.append(f"  lagged-coarse {k2}")
            Acur = Aci
    Aprev = Acur
    print(" | ".join(out), flush=True)
