"""CPU experiment (oracle only): coarse spaces built from RIGID-BODY modes.  A rigid motion exp(xi) of a piece of the world
(poses of an aggregate + the landmarks only they see) leaves every residual inside the piece unchanged; in the left-perturbation
chart of the poses (T <- exp(d) T, world -> camera) it reads d_i = Ad(T_i) xi -- NOT a constant d.  Coarse basis per aggregate:
the 6 columns of Ad(T_i) (and optionally the same times a linear weight), against the constant (+ linear) functions in the
chart that round 1 used."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
from scipy.spatial.transform import Rotation
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
    q, t, _ = o.state()
    return sp.bsr_matrix((data[order], c_all[order], indptr), shape=(n, n)).tocsr(), o.array("bsc").copy(), q[:P], t[:P]

def pcg(A, b, Minv, tol=1e-7, maxit=3000):
    x = np.zeros_like(b); r = b.copy(); z = Minv(r); p = z.copy(); rz = r @ z; rz0 = rz; k = 0
    while k < maxit and rz > tol * tol * rz0:
        q = A @ p; a = rz / (p @ q); x += a * p; r -= a * q; z = Minv(r); rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn; k += 1
    return k

def skew(v):
    K = np.zeros((len(v), 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -v[:, 2], v[:, 1]
    K[:, 1, 0], K[:, 1, 2] = v[:, 2], -v[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -v[:, 1], v[:, 0]
    return K

def adjoints(q, t, centre=None):
    """Ad(T_i) for [omega; upsilon] ordering; optional world-frame centre c (twists about c instead of the world origin:
    same span, better conditioned coarse matrix)."""
    R = Rotation.from_quat(q).as_matrix()
    tt = t if centre is None else t + np.einsum("nij,nj->ni", R, centre)
    Ad = np.zeros((len(q), 6, 6))
    Ad[:, :3, :3] = R; Ad[:, 3:, 3:] = R; Ad[:, 3:, :3] = skew(tt) @ R
    return Ad

idx = np.arange(P)
def prolong_blocks(J, blocks_list):
    """blocks_list: list of [P,6,6] arrays; coarse node of pose i for list entry m is len(list)*J[i]+m"""
    M = len(blocks_list)
    rows, cols, vals = [], [], []
    for m, B in enumerate(blocks_list):
        for a in range(6):
            for c in range(6):
                rows.append(6 * idx + a); cols.append(6 * (M * J + m) + c); vals.append(B[:, a, c])
    nn = M * (int(J.max()) + 1)
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, 6 * nn))

its = [] if len(sys.argv) > 3 else [int(a) for a in sys.argv[2].split(',')] if len(sys.argv) > 2 else [3, 9]
for it in its:
    A, b, q, t = system(it)
    Dinv = np.linalg.inv(np.stack([A[6*j:6*j+6, 6*j:6*j+6].toarray() for j in range(P)]))
    jac = lambda r: np.einsum("nij,nj->ni", Dinv, r.reshape(P, 6)).ravel()
    I6 = np.tile(np.eye(6), (P, 1, 1))
    R = Rotation.from_quat(q).as_matrix()
    C = -np.einsum("nji,nj->ni", R, t)          # camera centres in the world
    for g in (8, 12, 16, 24, 32, 48):
        J = idx // g
        mid = J * g + (np.minimum((J + 1) * g, P) - J * g - 1) / 2.0
        u = ((idx - mid) / (g / 2.0))[:, None, None]
        cen = np.stack([C[J == j].mean(0) for j in range(J.max() + 1)])[J]     # aggregate centroid
        Ad = np.stack([adjoints(q[i:i+1], t[i:i+1], cen[i])[0] for i in range(P)]) if False else None
        # vectorised: twists about each pose's aggregate centroid
        tt = t + np.einsum("nij,nj->ni", R, cen)
        Ad = np.zeros((P, 6, 6)); Ad[:, :3, :3] = R; Ad[:, 3:, 3:] = R; Ad[:, 3:, :3] = skew(tt) @ R
        for name, blocks in (("chart constants", [I6]), ("chart const + linear", [I6, I6 * u]),
                             ("rigid modes", [Ad]), ("rigid + linear rigid", [Ad, Ad * u])):
            Pm = prolong_blocks(J, blocks)
            Ac = (Pm.T @ A @ Pm).toarray()
            Aci = np.linalg.inv(Ac)
            k = pcg(A, b, lambda r: jac(r) + Pm @ (Aci @ (Pm.T @ r)))
            print(f"LM it {it}  agg {g:3d}  {name:22s} Nc {Ac.shape[0]:5d}  iterations {k:4d}  cond(Ac) {np.linalg.cond(Ac):.1e}", flush=True)

if len(sys.argv) > 3 and sys.argv[3] == "totals":
    tot = {}
    for it in range(10):
        A, b, q, t = system(it)
        Dinv = np.linalg.inv(np.stack([A[6*j:6*j+6, 6*j:6*j+6].toarray() for j in range(P)]))
        jac = lambda r: np.einsum("nij,nj->ni", Dinv, r.reshape(P, 6)).ravel()
        Dblk = sp.block_diag([Dinv[j] for j in range(P)], format="csr")
        I6 = np.tile(np.eye(6), (P, 1, 1))
        R = Rotation.from_quat(q).as_matrix()
        C = -np.einsum("nji,nj->ni", R, t)
        for g, kinds in ((24, ("chart c+l", "rigid+lin")), (12, ("rigid",)), (16, ("rigid+lin",)), (20, ("rigid+lin",))):
            J = idx // g
            mid = J * g + (np.minimum((J + 1) * g, P) - J * g - 1) / 2.0
            u = ((idx - mid) / (g / 2.0))[:, None, None]
            cen = np.stack([C[J == j].mean(0) for j in range(J.max() + 1)])[J]
            tt = t + np.einsum("nij,nj->ni", R, cen)
            Ad = np.zeros((P, 6, 6)); Ad[:, :3, :3] = R; Ad[:, 3:, 3:] = R; Ad[:, 3:, :3] = skew(tt) @ R
            for kind in kinds:
                blocks = {"chart c+l": [I6, I6 * u], "rigid": [Ad], "rigid+lin": [Ad, Ad * u]}[kind]
                Pm = prolong_blocks(J, blocks)
                for om in (0.0, 0.6):
                    Pv = Pm if om == 0 else (Pm - om * (Dblk @ (A @ Pm))).tocsr()
                    Aci = np.linalg.inv((Pv.T @ A @ Pv).toarray())
                    k = pcg(A, b, lambda r: jac(r) + Pv @ (Aci @ (Pv.T @ r)))
                    key = f"agg {g:2d} {kind:10s} smoothed {om}  Nc {Aci.shape[0]}"
                    tot.setdefault(key, []).append(k)
        print(it, {k: v[-1] for k, v in tot.items()}, flush=True)
    for k, v in tot.items():
        print(f"{k:50s} total {sum(v):5d}  per LM iteration {v}")
