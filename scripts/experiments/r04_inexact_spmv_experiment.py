"""CPU experiment (oracle only), round 4: PCG on the reduced system with an INEXACT matrix-vector product -- the matrix rounded to fp32
(what a half-size copy for the bandwidth-bound SpMV of the large shapes would be), exact fp64 vectors and accumulation -- from the
iteration at which the preconditioned residual has dropped by a given factor (relaxation strategy of inexact Krylov methods: the
admissible matvec error grows like 1 / ||r_k||).  Reported: iterations to the usual stop test (recurrence residual), the TRUE relative
residual and the relative error of the solution against a direct solve."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from oracle.oracle import OracleSolver
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
shape = sys.argv[1] if len(sys.argv) > 1 else "kitti00"
G = int(sys.argv[3]) if len(sys.argv) > 3 else {"kitti07": 8, "kitti00": 16, "s2m": 32, "g4m": 56}[shape]
fp = flatten(synth_named(shape))
o = OracleSolver(fp, RK, threads=16); o.build_structure()
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

def prolongation(g):
    idx = np.arange(P); J = idx // g; mid = J * g + (np.minimum((J + 1) * g, P) - J * g - 1) / 2.0
    w = (idx - mid) / (g / 2.0)
    rows, cols, vals = [], [], []
    for c in range(6):
        rows += [6 * idx + c, 6 * idx + c]; cols += [12 * J + c, 12 * J + 6 + c]; vals += [np.ones(P), w]
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, 12 * (J.max() + 1)))

Pm = prolongation(G)
def pcg(A, A32, b, Minv, switch, tol=1e-7, maxit=5000):
    x = np.zeros_like(b); r = b.copy(); z = Minv(r); p = z.copy(); rz = r @ z; rz0 = rz; k = 0; k32 = 0
    while k < maxit and rz > tol * tol * rz0:
        use32 = rz <= switch * switch * rz0
        q = (A32 if use32 else A) @ p; k32 += use32
        a = rz / (p @ q); x += a * p; r -= a * q; z = Minv(r); rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn; k += 1
    return x, k, k32

for it in ([int(a) for a in sys.argv[2].split(',')] if len(sys.argv) > 2 else [3, 9]):
    A, b = system(it)
    A32 = A.copy(); A32.data = A32.data.astype(np.float32).astype(np.float64)
    Dinv = np.linalg.inv(np.stack([A[6*j:6*j+6, 6*j:6*j+6].toarray() for j in range(P)]))
    Aci = np.linalg.inv((Pm.T @ A @ Pm).toarray())
    Minv = lambda r: np.einsum("nij,nj->ni", Dinv, r.reshape(P, 6)).ravel() + Pm @ (Aci @ (Pm.T @ r))
    xs = spla.spsolve(A.tocsc(), b)
    for switch in (0.0, 1e-2, 1e-1, 1.0, 2.0):
        x, k, k32 = pcg(A, A32, b, Minv, switch)
        print(f"{shape} LM iteration {it}: fp32 matrix once ||r||_M <= {switch:g} ||r0||_M: {k} iterations ({k32} with the fp32 matrix), "
              f"true residual {np.linalg.norm(b - A @ x) / np.linalg.norm(b):.2e}, solution error {np.linalg.norm(x - xs) / np.linalg.norm(xs):.2e}", flush=True)
