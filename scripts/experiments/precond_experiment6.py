"""CPU experiment (oracle only): does a bandwidth-reducing pose ORDER (reverse Cuthill-McKee on the block pattern of Hsc) help the
two-level preconditioner, whose aggregates are runs of consecutive pose indices?  Iterations (tol 1e-7) on the KITTI-00-shaped
reduced system for: the id order (= trajectory order), RCM of it, a random shuffle, RCM of the shuffle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee
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
    pat = sp.csr_matrix((np.ones(len(c_all)), c_all[order], indptr), shape=(P, P))
    return sp.bsr_matrix((data[order], c_all[order], indptr), shape=(n, n)).tocsr(), o.array("bsc").copy(), pat

def pcg(A, b, Minv, tol=1e-7, maxit=5000):
    x = np.zeros_like(b); r = b.copy(); z = Minv(r); p = z.copy(); rz = r @ z; rz0 = rz; k = 0
    while k < maxit and rz > tol * tol * rz0:
        q = A @ p; a = rz / (p @ q); x += a * p; r -= a * q; z = Minv(r); rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn; k += 1
    return k

def two_level_iterations(A, b, g=24):
    idx = np.arange(P); J = idx // g
    mid = J * g + (np.minimum((J + 1) * g, P) - J * g - 1) / 2.0
    rows, cols, vals = [], [], []
    for ni, w in ((2 * J, np.ones(P)), (2 * J + 1, (idx - mid) / (g / 2.0))):
        for c in range(6):
            rows.append(6 * idx + c); cols.append(6 * ni + c); vals.append(w)
    Pm = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, 12 * (J.max() + 1)))
    Aci = np.linalg.inv((Pm.T @ A @ Pm).toarray())
    Dinv = np.linalg.inv(np.stack([A[6*j:6*j+6, 6*j:6*j+6].toarray() for j in range(P)]))
    jac = lambda r: np.einsum("nij,nj->ni", Dinv, r.reshape(P, 6)).ravel()
    return pcg(A, b, lambda r: jac(r) + Pm @ (Aci @ (Pm.T @ r)))

def permuted(A, b, perm):
    """perm[new] = old pose"""
    idx = (6 * perm[:, None] + np.arange(6)[None, :]).ravel()
    return A[idx][:, idx].tocsr(), b[idx]

def band(pat, perm):
    inv = np.empty(P, dtype=np.int64); inv[perm] = np.arange(P)
    coo = pat.tocoo(); d = np.abs(inv[coo.row] - inv[coo.col])
    return int(d.max()), round(float(d.mean()), 1), round(float((d > 64).mean()), 3)

rng = np.random.default_rng(0)
for it in (3, 9):
    A, b, pat = system(it)
    ident = np.arange(P)
    shuf = np.concatenate([[0], 1 + rng.permutation(P - 1)])
    rcm_id = np.asarray(reverse_cuthill_mckee(pat, symmetric_mode=True))
    pat_s = pat[shuf][:, shuf].tocsr()
    rcm_sh = shuf[np.asarray(reverse_cuthill_mckee(pat_s, symmetric_mode=True))]
    for name, perm in (("id order", ident), ("RCM of id order", rcm_id), ("shuffled", shuf), ("RCM of shuffled", rcm_sh)):
        Ap, bp = permuted(A, b, perm)
        print(f"LM it {it}  {name:18s} band max/mean/frac>64 {band(pat, perm)}  iterations {two_level_iterations(Ap, bp)}", flush=True)

# ---- strongest-neighbour chain order from co-visibility counts (= Schur products per block) -------------------------------
def chain_order(W):
    """W: symmetric CSR of co-visibility weights (zero diagonal).  Greedy walk: start at the pose of smallest weighted degree,
    always step to the heaviest unvisited neighbour; when stuck, continue from the unvisited pose most strongly tied to the
    visited set (lazy heap)."""
    import heapq
    n = W.shape[0]
    indptr, indices, data = W.indptr, W.indices, W.data
    deg = np.asarray(W.sum(1)).ravel()
    visited = np.zeros(n, bool); order = []
    heap = []          # (-weight, pose) of unvisited poses adjacent to visited ones
    cur = int(np.argmin(np.where(deg > 0, deg, np.inf)))
    while len(order) < n:
        visited[cur] = True; order.append(cur)
        best, bw = -1, -1.0
        for k in range(indptr[cur], indptr[cur + 1]):
            j = indices[k]
            if not visited[j]:
                heapq.heappush(heap, (-data[k], j))
                if data[k] > bw: best, bw = j, data[k]
        if best >= 0: cur = best; continue
        cur = -1
        while heap:
            w, j = heapq.heappop(heap)
            if not visited[j]: cur = j; break
        if cur < 0:
            rest = np.nonzero(~visited)[0]
            if len(rest) == 0: break
            cur = int(rest[0])
    return np.array(order)

B = sp.csr_matrix((np.ones(fp.E), (fp.eP, fp.eL)), shape=(fp.Pt, fp.Lt))[:P, :fp.Lf]
W = (B @ B.T).tocsr(); W.setdiag(0); W.eliminate_zeros()
for it in (3, 9):
    A, b, pat = system(it)
    shuf = np.concatenate([[0], 1 + rng.permutation(P - 1)])
    Ws = W[shuf][:, shuf].tocsr()
    for name, perm in (("chain of id order", chain_order(W)), ("chain of shuffled", shuf[chain_order(Ws)])):
        Ap, bp = permuted(A, b, perm)
        print(f"LM it {it}  {name:18s} band max/mean/frac>64 {band(pat, perm)}  iterations {two_level_iterations(Ap, bp)}", flush=True)
