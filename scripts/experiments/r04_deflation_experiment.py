"""CPU experiment (oracle only), round 4: does a handful of GLOBAL vectors beside the aggregate coarse space cut the PCG iterations?
The lowest eigenvectors of the two-level-preconditioned reduced matrix M^-1 A (dense eigen-decomposition, KITTI-00 shape) are added as
extra columns of P -- fresh (from the same LM iteration) and stale (from an earlier LM iteration, what a GPU implementation that
harvests them from an earlier solve would have)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp, scipy.linalg as sla
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from oracle.oracle import OracleSolver
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
shape = sys.argv[1] if len(sys.argv) > 1 else "kitti00"
fp = flatten(synth_named(shape))
o = OracleSolver(fp, RK, threads=8); o.build_structure()
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
    return sp.bsr_matrix((data[order], c_all[order], indptr), shape=(n, n)).tocsr(), o.array("bsc").copy()

def pcg(A, b, Minv, tol=1e-7, maxit=3000):
    x = np.zeros_like(b); r = b.copy(); z = Minv(r); p = z.copy(); rz = r @ z; rz0 = rz; k = 0
    while k < maxit and rz > tol * tol * rz0:
        q = A @ p; a = rz / (p @ q); x += a * p; r -= a * q; z = Minv(r); rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn; k += 1
    return k

def analytic(g):
    idx = np.arange(P); J = idx // g; mid = J * g + (np.minimum((J + 1) * g, P) - J * g - 1) / 2.0
    w = (idx - mid) / (g / 2.0)
    rows, cols, vals = [], [], []
    for c in range(6):
        rows += [6 * idx + c, 6 * idx + c]; cols += [12 * J + c, 12 * J + 6 + c]; vals += [np.ones(P), w]
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, 12 * (J.max() + 1))).toarray()

G = 16 if shape == "kitti00" else 8
Pm = analytic(G)
def two_level(A, Pext):
    Dinv = np.linalg.inv(np.stack([A[6*j:6*j+6, 6*j:6*j+6].toarray() for j in range(P)]))
    Aci = np.linalg.inv(Pext.T @ (A @ Pext))
    return Dinv, Aci
def apply(Dinv, Aci, Pext):
    return lambda r: np.einsum("nij,nj->ni", Dinv, r.reshape(P, 6)).ravel() + Pext @ (Aci @ (Pext.T @ r))

def low_modes(A, k):
    """k lowest eigenvectors of M^-1 A for the plain two-level M (generalised symmetric problem A v = mu M v)"""
    Dinv, Aci = two_level(A, Pm)
    Minv = sla.block_diag(*Dinv) + Pm @ Aci @ Pm.T
    Ad = A.toarray()
    t = time.time()
    # A v = mu M v  <=>  (Minv-symmetrised): eigh(A, M) with M = inv(Minv); use eigh(Minv^{1/2} A Minv^{1/2}) through Cholesky of Minv
    L = np.linalg.cholesky(Minv)
    w, Y = sla.eigh(L.T @ Ad @ L, subset_by_index=[0, k - 1])
    V = L @ Y
    print(f"   (dense eigen-decomposition {time.time() - t:.0f} s; lowest eigenvalues of M^-1 A: {np.round(w[:8], 4).tolist()})", flush=True)
    return V

its = [int(a) for a in sys.argv[2].split(',')] if len(sys.argv) > 2 else [3, 9]
stale = None
for it in its:
    A, b = system(it)
    Dinv, Aci = two_level(A, Pm)
    base = pcg(A, b, apply(Dinv, Aci, Pm))
    print(f"LM iteration {it}: two-level as on the GPU (aggregates of {G}, coarse dim {Pm.shape[1]}): {base} iterations", flush=True)
    V = low_modes(A, 40)
    for k in (10, 20, 40):
        Pe = np.hstack([Pm, V[:, :k]]); D2, A2 = two_level(A, Pe)
        print(f"LM iteration {it}:   + {k:2d} FRESH lowest eigenvectors of M^-1 A as global coarse functions: {pcg(A, b, apply(D2, A2, Pe))} iterations", flush=True)
        if stale is not None:
            Pe = np.hstack([Pm, stale[1][:, :k]]); D2, A2 = two_level(A, Pe)
            print(f"LM iteration {it}:   + {k:2d} STALE ones (from LM iteration {stale[0]}): {pcg(A, b, apply(D2, A2, Pe))} iterations", flush=True)
    if stale is None or "--latest" in sys.argv: stale = (it, V)
