"""CPU experiment (oracle only), round 4: PCG iterations (tol 1e-7) of the two-level preconditioner on the KITTI-00-shaped reduced system
with a SPECTRAL coarse space -- per aggregate the k lowest eigenvectors of the aggregate's own diagonal block of the block-Jacobi-scaled
matrix (a GenEO-like local basis) -- against the analytic constant + linear basis the GPU path uses, at equal coarse dimension."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp, scipy.linalg as sla
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from oracle.oracle import OracleSolver
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
fp = flatten(synth_named(sys.argv[1] if len(sys.argv) > 1 else "kitti00"))
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
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, 12 * (J.max() + 1)))

def spectral(A, Dinv_sqrt, g, k, neumann):
    """k lowest eigenvectors per aggregate of the symmetrically Jacobi-scaled local block (Dirichlet: the diagonal block itself;
    'neumann': the block with the couplings to other aggregates folded onto its diagonal blocks' row sums left out -- here simply the
    block assembled from intra-aggregate couplings only, diagonal blocks reduced by the absolute row sums of what leaves)"""
    cols = []
    nagg = (P + g - 1) // g
    data = []
    for J in range(nagg):
        lo, hi = 6 * J * g, 6 * min(P, (J + 1) * g)
        B = A[lo:hi, lo:hi].toarray()
        if neumann:
            out = A[lo:hi, :].copy().tolil(); out[:, lo:hi] = 0; out = sp.csr_matrix(out)
            B = B - np.diag(np.asarray(abs(out).sum(axis=1)).ravel())
            B = 0.5 * (B + B.T)
        S = Dinv_sqrt[lo:hi]
        w, V = sla.eigh(S[:, None] * B * S[None, :])
        kk = min(k, hi - lo)
        V = S[:, None] * V[:, :kk]
        data.append((lo, hi, V))
    tot = sum(d[2].shape[1] for d in data)
    Pm = sp.lil_matrix((n, tot)); c0 = 0
    for lo, hi, V in data:
        Pm[lo:hi, c0:c0 + V.shape[1]] = V; c0 += V.shape[1]
    return Pm.tocsr()

for it in ((3, 6, 9) if len(sys.argv) < 3 else [int(a) for a in sys.argv[2].split(',')]):
    A, b = system(it)
    Dblocks = np.stack([A[6*j:6*j+6, 6*j:6*j+6].toarray() for j in range(P)])
    Dinv = np.linalg.inv(Dblocks)
    jac = lambda r: np.einsum("nij,nj->ni", Dinv, r.reshape(P, 6)).ravel()
    dsq = 1.0 / np.sqrt(A.diagonal())
    cases = [("constant + linear, 16 poses/aggregate (GPU default)", analytic(16)), ("constant + linear, 24 poses/aggregate", analytic(24))]
    for g, k in ((16, 12), (24, 18), (32, 24), (16, 8), (24, 12)):
        cases.append((f"spectral (Dirichlet block), {g} poses/aggregate, {k} vectors", spectral(A, dsq, g, k, False)))
        cases.append((f"spectral (row-sum-reduced block), {g} poses/aggregate, {k} vectors", spectral(A, dsq, g, k, True)))
    for nm, Pv in cases:
        Ac = (Pv.T @ A @ Pv).toarray()
        Aci = np.linalg.inv(Ac)
        kk = pcg(A, b, lambda r: jac(r) + Pv @ (Aci @ (Pv.T @ r)))
        print(f"LM iteration {it}: {nm:75s} coarse dim {Ac.shape[0]:5d}  iterations {kk}", flush=True)
