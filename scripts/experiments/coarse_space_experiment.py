"""CPU experiment (oracle only): PCG iterations (tol 1e-7) of the two-level preconditioner on the KITTI-00-shaped reduced
system for different coarse spaces of (nearly) the same dimension: piecewise constant per aggregate (what the GPU path
uses), constant + linear per aggregate, and linear hat functions between aggregate centres."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
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

def pcg(A, b, Minv, tol=1e-7, maxit=5000):
    x = np.zeros_like(b); r = b.copy(); z = Minv(r); p = z.copy(); rz = r @ z; rz0 = rz; k = 0
    while k < maxit and rz > tol * tol * rz0:
        q = A @ p; a = rz / (p @ q); x += a * p; r -= a * q; z = Minv(r); rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn; k += 1
    return k

def prolong(weights):
    """weights: list of (pose index array, coarse node index array, weight array) -> n x 6*nodes matrix (identity on the 6 dof)"""
    rows, cols, vals = [], [], []
    for pi, ni, w in weights:
        for c in range(6):
            rows.append(6 * pi + c); cols.append(6 * ni + c); vals.append(w)
    nn = max(int(ni.max()) for _, ni, _ in weights) + 1
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, 6 * nn))

idx = np.arange(P)
def spaces():
    for g in (12,):
        yield f"constant, {g} poses/aggregate", prolong([(idx, idx // g, np.ones(P))])
    for g in (24, 32):
        J = idx // g; mid = J * g + (np.minimum((J + 1) * g, P) - J * g - 1) / 2.0
        yield f"constant + linear, {g} poses/aggregate", prolong([(idx, 2 * J, np.ones(P)), (idx, 2 * J + 1, (idx - mid) / (g / 2.0))])
    for g in (36, 48):
        J = idx // g; mid = J * g + (np.minimum((J + 1) * g, P) - J * g - 1) / 2.0
        u = (idx - mid) / (g / 2.0)
        yield f"constant + linear + quadratic, {g} poses/aggregate", prolong([(idx, 3 * J, np.ones(P)), (idx, 3 * J + 1, u), (idx, 3 * J + 2, 1.5 * u * u - 0.5)])
    for g in (12, 16, 24):
        s = idx / g; left = np.floor(s).astype(int); w = s - left
        yield f"linear hats, nodes every {g} poses", prolong([(idx, left, 1 - w), (idx, left + 1, w)])

for it in ((3, 6, 9) if len(sys.argv) < 3 else [int(a) for a in sys.argv[2].split(',')]):
    A, b = system(it)
    Dinv = np.linalg.inv(np.stack([A[6*j:6*j+6, 6*j:6*j+6].toarray() for j in range(P)]))
    jac = lambda r: np.einsum("nij,nj->ni", Dinv, r.reshape(P, 6)).ravel()
    Dblk = sp.block_diag([Dinv[j] for j in range(P)], format="csr")
    for name, Pm in spaces():
        variants = [(name, Pm)]
        if "constant + linear, 24" in name or "constant + linear, 32" in name:
            for om in (0.3, 0.6):       # smoothed prolongator (smoothed aggregation): P_s = (I - omega D^-1 A) P -- costs two more SpMVs per iteration
                variants.append((name + f", smoothed omega={om}", (Pm - om * (Dblk @ (A @ Pm))).tocsr()))
        for nm, Pv in variants:
            Ac = (Pv.T @ A @ Pv).toarray()
            Aci = np.linalg.inv(Ac)
            k = pcg(A, b, lambda r: jac(r) + Pv @ (Aci @ (Pv.T @ r)))
            print(f"LM iteration {it}: {nm:62s} coarse dim {Ac.shape[0]:5d}  iterations {k}", flush=True)
