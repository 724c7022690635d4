"""CPU experiment (oracle only): three-level additive preconditioner for the large shapes.  Level 1 = aggregates of 24 poses with
constant + linear functions (as at KITTI-00); its coarse matrix is too large to invert densely at 5-10 k poses, so it is smoothed
by its own 12x12 block-Jacobi and corrected by a level 2 of groups of level-1 nodes (dense inverse).  Compared with the current
two-level scheme whose aggregates grow with the pose count."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from oracle.oracle import OracleSolver
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
shape = sys.argv[1] if len(sys.argv) > 1 else "s2m"
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

def prolong_cl(nfine, comps, g):
    """constant + linear per aggregate of g consecutive fine nodes, `comps` components per node"""
    idx = np.arange(nfine); J = idx // g
    mid = J * g + (np.minimum((J + 1) * g, nfine) - J * g - 1) / 2.0
    rows, cols, vals = [], [], []
    for a, w in ((0, np.ones(nfine)), (1, (idx - mid) / (g / 2.0))):
        for c in range(comps):
            rows.append(comps * idx + c); cols.append(2 * comps * J + comps * a + c); vals.append(w)
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(comps * nfine, 2 * comps * (J.max() + 1)))

def prolong_const(nfine, comps, g):
    idx = np.arange(nfine); J = idx // g
    rows = (comps * idx[:, None] + np.arange(comps)[None, :]).ravel(); cols = (comps * J[:, None] + np.arange(comps)[None, :]).ravel()
    return sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(comps * nfine, comps * (J.max() + 1)))

def block_diag_inv(M, bs):
    nb = M.shape[0] // bs
    blocks = np.stack([M[bs*j:bs*j+bs, bs*j:bs*j+bs].toarray() for j in range(nb)])
    inv = np.linalg.inv(blocks)
    return lambda r: np.einsum("nij,nj->ni", inv, r.reshape(nb, bs)).ravel()

for it in [int(a) for a in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["9"])]:
    A, b = system(it)
    jac = block_diag_inv(A, 6)
    auto = max(24, (P + 114) // 115)
    P1 = prolong_cl(P, 6, auto); Ac = (P1.T @ A @ P1).toarray(); Aci = np.linalg.inv(Ac)
    print(f"{shape} LM it {it}: two-level, agg {auto}, Nc {Ac.shape[0]}: {pcg(A, b, lambda r: jac(r) + P1 @ (Aci @ (P1.T @ r)))} iterations", flush=True)
    P1 = prolong_cl(P, 6, 24); A1 = (P1.T @ A @ P1).tocsr(); n1 = A1.shape[0] // 12
    lu = spl.splu(A1.tocsc())
    print(f"   level 1 = 24 poses, Nc1 {A1.shape[0]}, EXACT level-1 solve (lower bound): {pcg(A, b, lambda r: jac(r) + P1 @ lu.solve(P1.T @ r))} iterations", flush=True)
    j1 = block_diag_inv(A1, 12)
    for g2, kind in ((4, "const"), (8, "const"), (8, "c+l"), (16, "c+l")):
        P2 = prolong_const(n1, 12, g2) if kind == "const" else prolong_cl(n1, 12, g2)
        A2i = np.linalg.inv((P2.T @ A1 @ P2).toarray())
        def M1(r1): return j1(r1) + P2 @ (A2i @ (P2.T @ r1))
        k = pcg(A, b, lambda r: jac(r) + P1 @ M1(P1.T @ r))
        print(f"   three-level additive: level 2 = {g2} level-1 nodes ({kind}), Nc2 {A2i.shape[0]}: {k} iterations", flush=True)
