"""CPU experiment (oracle only, no GPU): can the dense coarse inverse of the two-level preconditioner be carried from
one LM trial to the next by Newton-Schulz refinement X <- X (2I - Ac X) instead of a fresh Gauss-Jordan inversion?
For every LM iteration of a 10-iteration run: ||I - Ac_i X_{i-1}||_2, and PCG iterations (tol 1e-7) with the exact
inverse, with the stale inverse and after 1 / 2 refinement steps.   python scripts/experiments/ns_experiment.py [shape] [agg]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from oracle.oracle import OracleSolver

RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
name = sys.argv[1] if len(sys.argv) > 1 else "kitti00"
g = int(sys.argv[2]) if len(sys.argv) > 2 else 16
fp = flatten(synth_named(name))
o = OracleSolver(fp, RK); o.build_structure()
q0, t0, X0 = o.state()
lams = o.optimize(10)["lambdas"]
P = fp.Pf; n = 6 * P
nb = (P + g - 1) // g
agg = np.arange(P) // g
Pm = sp.csr_matrix((np.ones(n), (np.arange(n), 6 * np.repeat(agg, 6) + np.tile(np.arange(6), P))), shape=(n, 6 * nb))

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
    A = sp.bsr_matrix((data[order], c_all[order], indptr), shape=(n, n)).tocsr()
    return A, o.array("bsc").copy()

def pcg(A, b, Minv, tol=1e-7, maxit=5000):
    x = np.zeros_like(b); r = b.copy(); z = Minv(r); p = z.copy(); rz = r @ z; rz0 = rz; k = 0
    while k < maxit and rz > tol * tol * rz0:
        q = A @ p; a = rz / (p @ q); x += a * p; r -= a * q; z = Minv(r); rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn; k += 1
    return k

Xns = None
tot = dict(exact=0, stale=0, ns1=0, ns2=0)
for i in range(len(lams)):
    A, b = system(i)
    Dinv = np.linalg.inv(np.stack([A[6*j:6*j+6, 6*j:6*j+6].toarray() for j in range(P)]))
    jac = lambda r: np.einsum("nij,nj->ni", Dinv, r.reshape(P, 6)).ravel()
    Ac = (Pm.T @ A @ Pm).toarray()
    Aci = np.linalg.inv(Ac)
    two = lambda X: (lambda r: jac(r) + Pm @ (X @ (Pm.T @ r)))
    k_exact = pcg(A, b, two(Aci)); tot["exact"] += k_exact
    if Xns is None:
        Xns = Aci.copy(); print(f"it {i} lambda {lams[i]:.3e}: exact {k_exact} (first solve: Gauss-Jordan)")
        for key in ("stale", "ns1", "ns2"): tot[key] += k_exact
        X1 = X2 = Aci
        continue
    rho = np.linalg.norm(np.eye(6 * nb) - Ac @ X1, 2)
    k_stale = pcg(A, b, two(X1))
    X1n = X1 @ (2 * np.eye(6 * nb) - Ac @ X1); X1n = 0.5 * (X1n + X1n.T)
    k1 = pcg(A, b, two(X1n))
    rho2 = np.linalg.norm(np.eye(6 * nb) - Ac @ X2, 2)
    X2n = X2
    for _ in range(2): X2n = X2n @ (2 * np.eye(6 * nb) - Ac @ X2n)
    X2n = 0.5 * (X2n + X2n.T)
    k2 = pcg(A, b, two(X2n))
    print(f"it {i} lambda {lams[i]:.3e}: exact {k_exact} | chain of 1-step refinements: ||I-AcX||={rho:.3f} stale {k_stale} refined {k1} "
          f"(min eig {np.linalg.eigvalsh(X1n).min():.2e}) | chain of 2-step: ||I-AcX||={rho2:.3f} refined {k2}", flush=True)
    tot["stale"] += k_stale; tot["ns1"] += k1; tot["ns2"] += k2
    X1, X2 = X1n, X2n
print("totals", tot)
