"""CPU experiment 3: block size of the Jacobi level (1/2/4/8 poses per block) combined with the aggregate coarse level."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from oracle.oracle import OracleSolver
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
fp = flatten(synth_named("kitti00")); P = fp.Pf; n = 6 * P
def system(nit):
    o = OracleSolver(fp, RK); res = o.optimize(nit)
    o.compute_errors(); o.build_system(); lam = res["lambdas"][-1]; o.set_lambda(lam); o.schur()
    rp, ci, v = o.hsc(); b = o.array("bsc")
    rows = np.repeat(np.arange(P), np.diff(rp))
    data = np.concatenate([v, v[rows != ci].transpose(0, 2, 1)])
    r_all = np.concatenate([rows, ci[rows != ci]]); c_all = np.concatenate([ci, rows[rows != ci]])
    order = np.lexsort((c_all, r_all)); indptr = np.concatenate([[0], np.cumsum(np.bincount(r_all, minlength=P))])
    return sp.bsr_matrix((data[order], c_all[order], indptr), shape=(n, n)).tocsr(), b, lam
def pcg(A, b, Minv, tol=1e-8, maxit=5000):
    x = np.zeros_like(b); r = b.copy(); z = Minv(r); p = z.copy(); rz = r @ z; rz0 = rz; k = 0
    while k < maxit and rz > tol * tol * rz0:
        q = A @ p; a = rz / (p @ q); x += a * p; r -= a * q; z = Minv(r); rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn; k += 1
    return k
for nit in (3, 6, 9):
    A, b, lam = system(nit)
    out = [f"LM it {nit} lam {lam:.3g}"]
    g = 16; nb = (P + g - 1) // g; agg = np.arange(P) // g
    Pm = sp.csr_matrix((np.ones(n), (np.arange(n), 6 * np.repeat(agg, 6) + np.tile(np.arange(6), P))), shape=(n, 6 * nb))
    Aci = np.linalg.inv((Pm.T @ A @ Pm).toarray())
    for bs in (1, 2, 4, 8):
        blocks = []
        for I in range((P + bs - 1) // bs):
            s, e = 6 * bs * I, min(6 * bs * (I + 1), n); blocks.append((s, e, np.linalg.inv(A[s:e, s:e].toarray())))
        def bj(r):
            out_ = np.empty_like(r)
            for s, e, Bi in blocks: out_[s:e] = Bi @ r[s:e]
            return out_
        out.append(f"bs{bs}: jac {pcg(A, b, bj)} +coarse16 {pcg(A, b, lambda r: bj(r) + Pm @ (Aci @ (Pm.T @ r)))}")
    print(" | ".join(out), flush=True)
