"""CPU experiment (oracle only): PCG iterations (tol 1e-7) on the KITTI-00-shaped reduced system with stronger fine-level
parts of the two-level preconditioner: dense aggregate blocks (non-overlapping and overlapping additive Schwarz) next to the
6x6 block-Jacobi used so far, all with the constant + linear coarse space."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp, scipy.linalg as sl
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from oracle.oracle import OracleSolver
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
shape = sys.argv[1] if len(sys.argv) > 1 else "kitti00"
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
    return sp.bsr_matrix((data[order], c_all[order], indptr), shape=(n, n)).tocsr(), o.array("bsc").copy()

def pcg(A, b, Minv, tol=1e-7, maxit=3000):
    x = np.zeros_like(b); r = b.copy(); z = Minv(r); p = z.copy(); rz = r @ z; rz0 = rz; k = 0
    while k < maxit and rz > tol * tol * rz0:
        q = A @ p; a = rz / (p @ q); x += a * p; r -= a * q; z = Minv(r); rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn; k += 1
    return k

idx = np.arange(P)
def coarse(g):
    J = idx // g; mid = J * g + (np.minimum((J + 1) * g, P) - J * g - 1) / 2.0
    rows, cols, vals = [], [], []
    for pi, ni, w in [(idx, 2 * J, np.ones(P)), (idx, 2 * J + 1, (idx - mid) / (g / 2.0))]:
        for c in range(6):
            rows.append(6 * pi + c); cols.append(6 * ni + c); vals.append(w)
    nn = 2 * (int(J.max()) + 1)
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, 6 * nn))

def schwarz(A, size, overlap):
    """additive Schwarz with dense subdomain solves: subdomain s = poses [s*size - overlap, (s+1)*size + overlap)"""
    doms = []
    for s0 in range(0, P, size):
        lo, hi = max(0, s0 - overlap), min(P, s0 + size + overlap)
        sl_ = slice(6 * lo, 6 * hi)
        doms.append((sl_, sl.cho_factor(A[sl_, sl_].toarray())))
    def apply(r):
        z = np.zeros_like(r)
        for sl_, c in doms:
            z[sl_] += sl.cho_solve(c, r[sl_])
        return z
    return apply

its = [int(a) for a in sys.argv[2].split(',')] if len(sys.argv) > 2 else [3, 9]
for it in its:
    A, b = system(it)
    Dinv = np.linalg.inv(np.stack([A[6*j:6*j+6, 6*j:6*j+6].toarray() for j in range(P)]))
    jac = lambda r: np.einsum("nij,nj->ni", Dinv, r.reshape(P, 6)).ravel()
    for g in (24,) if shape == "kitti00" else (24, 48):
        Pm = coarse(g)
        Aci = np.linalg.inv((Pm.T @ A @ Pm).toarray())
        cc = lambda r: Pm @ (Aci @ (Pm.T @ r))
        print(f"LM it {it} agg {g} Nc {Aci.shape[0]}: BJ6 + coarse: {pcg(A, b, lambda r: jac(r) + cc(r))}", flush=True)
        for size, ov in ((4, 0), (8, 0), (12, 0), (24, 0), (12, 4), (24, 4), (24, 8), (24, 12), (48, 0)):
            S = schwarz(A, size, ov)
            k1 = pcg(A, b, lambda r: S(r) + cc(r))
            k0 = pcg(A, b, S)
            print(f"    dense blocks of {size} poses, overlap {ov}: with coarse {k1}   without coarse {k0}", flush=True)
