"""CPU experiment (oracle only): a THREE-level additive preconditioner for the large shapes, where the two-level one needs aggregates of
32-56 poses to keep its dense coarse inverse affordable and pays with 54-84 PCG iterations per solve.
  two-level (built)   M^-1 = D^-1 + P A_c^-1 P^T                                   aggregates of g poses, dense inverse of dimension 12 P/g
  three-level         M^-1 = D^-1 + P2 D2^-1 P2^T + P3 A_3^-1 P3^T                 small aggregates (g2 poses) with only their 12x12 diagonal blocks
                                                                                    inverted, dense inverse on aggregates of g2*g3 poses
Both use the constant + linear coarse functions of the built solver; span(P3) is contained in span(P2).
   python scripts/experiments/precond_experiment9.py s2m 3,9"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from oracle.oracle import OracleSolver
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
shape = sys.argv[1] if len(sys.argv) > 1 else "kitti00"
its = [int(a) for a in sys.argv[2].split(',')] if len(sys.argv) > 2 else [3, 9]
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

def blockdiag_inv(M, bs):
    nb = M.shape[0] // bs
    B = np.stack([M[bs * j:bs * j + bs, bs * j:bs * j + bs].toarray() for j in range(nb)])
    return np.linalg.inv(B)

auto = {"kitti07": 8, "kitti00": 16, "s2m": 32, "g4m": 56}.get(shape, 16)
for it in its:
    A, b = system(it)
    Dinv = blockdiag_inv(A, 6)
    jac = lambda r: np.einsum("nij,nj->ni", Dinv, r.reshape(P, 6)).ravel()
    Pm = coarse(auto); Aci = np.linalg.inv((Pm.T @ A @ Pm).toarray())
    base = pcg(A, b, lambda r: jac(r) + Pm @ (Aci @ (Pm.T @ r)))
    print(f"{shape} LM it {it}: two-level, aggregates of {auto} (Nc {Aci.shape[0]}): {base} iterations", flush=True)
    for g2, g3 in ((4, 8), (8, 4), (8, 8), (16, 4), (16, 8), (8, 16)):
        if g2 * g3 * 2 > P: continue
        P2 = coarse(g2); Ac2 = (P2.T @ A @ P2).tocsr()
        D2 = blockdiag_inv(Ac2, 12); nc2 = D2.shape[0]
        P3 = coarse(g2 * g3); A3i = np.linalg.inv((P3.T @ A @ P3).toarray())
        lvl2 = lambda r: P2 @ np.einsum("nij,nj->ni", D2, (P2.T @ r).reshape(nc2, 12)).ravel()
        lvl3 = lambda r: P3 @ (A3i @ (P3.T @ r))
        k_add = pcg(A, b, lambda r: jac(r) + lvl2(r) + lvl3(r))
        k_half = pcg(A, b, lambda r: jac(r) + 0.5 * lvl2(r) + lvl3(r))
        k_23 = pcg(A, b, lambda r: jac(r) + lvl3(r))
        print(f"    g2 {g2:2d} x g3 {g3:2d} (Nc2 {12 * nc2}, Nc3 {A3i.shape[0]}): additive 3-level {k_add}   level-2 term halved {k_half}   without level 2 {k_23}", flush=True)
