"""CPU experiment: PCG iteration counts on the KITTI-00-shaped reduced system for several preconditioners."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from oracle.oracle import OracleSolver

RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
name = sys.argv[1] if len(sys.argv) > 1 else "kitti00"
fp = flatten(synth_named(name))
o = OracleSolver(fp, RK)
nit = int(sys.argv[2]) if len(sys.argv) > 2 else 9
res = o.optimize(nit)
lam = res["lambdas"][-1]
o.compute_errors(); o.build_system(); o.set_lambda(lam); o.schur()
rp, ci, v = o.hsc()
b = o.array("bsc")
P = fp.Pf; n = 6 * P
rows = np.repeat(np.arange(P), np.diff(rp))
# full symmetric BSR
data = np.concatenate([v, v[rows != ci].transpose(0, 2, 1)])
r_all = np.concatenate([rows, ci[rows != ci]]); c_all = np.concatenate([ci, rows[rows != ci]])
order = np.lexsort((c_all, r_all))
indptr = np.concatenate([[0], np.cumsum(np.bincount(r_all, minlength=P))])
A = sp.bsr_matrix((data[order], c_all[order], indptr), shape=(n, n)).tocsr()
print(name, "lambda", lam, "n", n, "nnz blocks", len(order))
xref = spl.spsolve(A.tocsc(), b)

def pcg(A, b, Minv, tol=1e-10, maxit=20000):
    x = np.zeros_like(b); r = b.copy(); z = Minv(r); p = z.copy(); rz = r @ z; rz0 = rz; k = 0
    while k < maxit and rz > tol * tol * rz0:
        q = A @ p; a = rz / (p @ q); x += a * p; r -= a * q; z = Minv(r); rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn; k += 1
    return x, k

D = np.stack([A[6*i:6*i+6, 6*i:6*i+6].toarray() for i in range(P)])
Dinv = np.linalg.inv(D)
def jac(r): return np.einsum("nij,nj->ni", Dinv, r.reshape(P, 6)).ravel()
x, k = pcg(A, b, jac); print("block-Jacobi 6x6: iters", k, "err", np.linalg.norm(x - xref) / np.linalg.norm(xref))

for g in (2, 4, 8):
    nb = (P + g - 1) // g
    blocks = []
    for I in range(nb):
        s, e = 6 * g * I, min(6 * g * (I + 1), n)
        blocks.append(np.linalg.inv(A[s:e, s:e].toarray()))
    def bj(r, blocks=blocks, g=g):
        out = np.empty_like(r)
        for I, Bi in enumerate(blocks):
            s = 6 * g * I; e = s + Bi.shape[0]
            out[s:e] = Bi @ r[s:e]
        return out
    x, k = pcg(A, b, bj); print(f"block-Jacobi {g} poses/block: iters", k)

for g in (4, 8, 16, 32):
    nb = (P + g - 1) // g
    agg = np.arange(P) // g
    Pm = sp.csr_matrix((np.ones(n), (np.arange(n), 6 * np.repeat(agg, 6) + np.tile(np.arange(6), P))), shape=(n, 6 * nb))
    Ac = (Pm.T @ A @ Pm).toarray()
    Aci = np.linalg.inv(Ac)
    def two(r, Pm=Pm, Aci=Aci): return jac(r) + Pm @ (Aci @ (Pm.T @ r))
    x, k = pcg(A, b, two); print(f"two-level additive, aggregates of {g} poses (coarse dim {6*nb}): iters", k, "err", np.linalg.norm(x - xref) / np.linalg.norm(xref))

# overlapping additive Schwarz windows + coarse
for w, ov in ((16, 4), (32, 8)):
    starts = list(range(0, P, w - ov))
    wins = []
    for s0 in starts:
        s, e = 6 * s0, min(6 * (s0 + w), n)
        wins.append((s, e, np.linalg.inv(A[s:e, s:e].toarray())))
    g = 16; nb = (P + g - 1) // g; agg = np.arange(P) // g
    Pm = sp.csr_matrix((np.ones(n), (np.arange(n), 6 * np.repeat(agg, 6) + np.tile(np.arange(6), P))), shape=(n, 6 * nb))
    Aci = np.linalg.inv((Pm.T @ A @ Pm).toarray())
    def asm(r, wins=wins):
        out = np.zeros_like(r)
        for s, e, Bi in wins: out[s:e] += Bi @ r[s:e]
        return out
    x, k = pcg(A, b, asm); print(f"additive Schwarz w={w} ov={ov}: iters", k)
    x, k = pcg(A, b, lambda r: asm(r) + Pm @ (Aci @ (Pm.T @ r))); print(f"additive Schwarz w={w} ov={ov} + coarse(16): iters", k)
