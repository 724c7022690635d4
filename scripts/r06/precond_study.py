"""CPU study (numpy / scipy, the oracle builds the reduced systems): PCG iteration counts of the two-level preconditioner on a
KITTI-00-shaped graph for different smoother blocks -- 6 x 6 pose blocks (what the solver applies today) against dense blocks of
`sb` consecutive poses -- with the same coarse space (aggregates of `agg` poses, constant + linear functions)."""
import sys, os, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from oracle.oracle import OracleSolver

shape = sys.argv[1] if len(sys.argv) > 1 else "kitti00"
agg = int(sys.argv[2]) if len(sys.argv) > 2 else 24
fp = flatten(synth_named(shape))
rk = ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815))))
o = OracleSolver(fp, rk)
o.build_structure()

def system(lam):
    o.compute_errors(); o.build_system(); o.set_lambda(lam); o.schur()
    rp, ci, v = o.hsc()
    Pf = fp.Pf
    rows = np.repeat(np.arange(Pf), np.diff(rp))
    U = sp.bsr_matrix((v, ci, rp), shape=(6 * Pf, 6 * Pf)).tocsr()
    D = sp.bsr_matrix((v[rows == ci], np.arange(Pf), np.arange(Pf + 1)), shape=(6 * Pf, 6 * Pf)).tocsr()
    A = (U + U.T - D).tocsr()
    b = o.array("bsc") if "bsc" in o.ARR else np.ones(6 * Pf)
    o.restore_diagonal()
    return A, b[:6 * Pf]

def coarse(A, Pf, agg, cl=2):
    nc = (Pf + agg - 1) // agg
    r, c, w = [], [], []
    for i in range(Pf):
        J, il = divmod(i, agg)
        wl = 0.0 if (i == Pf - 1 and il == 0) else (2 * il + 1 - agg) / agg
        for k in range(6):
            r.append(6 * i + k); c.append(6 * cl * J + k); w.append(1.0)
            if cl == 2: r.append(6 * i + k); c.append(6 * cl * J + 6 + k); w.append(wl)
    P = sp.csr_matrix((w, (r, c)), shape=(6 * Pf, 6 * cl * nc))
    Ac = (P.T @ A @ P).toarray()
    return P, np.linalg.inv(Ac)

def smoother(A, Pf, sb):
    blocks = []
    for i0 in range(0, Pf, sb):
        i1 = min(Pf, i0 + sb)
        blocks.append(np.linalg.inv(A[6 * i0:6 * i1, 6 * i0:6 * i1].toarray()))
    return sp.block_diag(blocks).tocsr()

def pcg(A, b, M, tol=1e-7, maxit=500):
    x = np.zeros_like(b); r = b.copy(); z = M(r); p = z.copy(); rz = r @ z; rz0 = rz
    for k in range(maxit):
        if not rz > tol * tol * rz0: return k
        q = A @ p; alpha = rz / (p @ q)
        x += alpha * p; r -= alpha * q
        z = M(r); rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn
    return maxit

lam0 = None
for lam_scale in (1.0, 1e-1, 1e-2, 1e-3, 1e-4):
    if lam0 is None:
        o.compute_errors(); o.build_system(); lam0 = 1e-5 * o.max_diagonal()
    lam = lam0 * lam_scale
    A, b = system(lam)
    Pf = fp.Pf
    P, Aci = coarse(A, Pf, agg)
    out = []
    for sb in (1, 2, 4, 6, 8, 12, 24):
        S = smoother(A, Pf, sb)
        M = lambda r: S @ r + P @ (Aci @ (P.T @ r))
        out.append((sb, pcg(A, b, M)))
    # multiplicative variant with the 6x6 smoother (symmetric: smoother, coarse, smoother) for reference
    S1 = smoother(A, Pf, 1)
    def Mmul(r):
        z = S1 @ r; z = z + P @ (Aci @ (P.T @ (r - A @ z))); return z + S1 @ (r - A @ z)
    print(f"lambda {lam:.3e}: iterations by smoother block (poses): {out}; multiplicative 6x6: {pcg(A, b, Mmul)}", flush=True)
