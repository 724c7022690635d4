"""Pins for the oracle's per-edge / per-vertex math (the reference ships no tests: SURVEY.md section 4).
Each check is independent of the reference's formulas: finite differences, mpmath, numpy/scipy."""
import mpmath as mp
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from oracle import oracle as orc

CAM = np.array([718.856, 718.856, 607.1928, 185.2157, 386.1448])


def rand_pose(rng):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    if q[3] < 0:
        q = -q
    return q, rng.normal(size=3)


def se3_exp_np(d):
    """exp([w;u]) with mpmath-free numpy via the matrix exponential series (independent of Rodrigues code)."""
    w, u = d[:3], d[3:]
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    T = np.zeros((4, 4)); T[:3, :3] = K; T[:3, 3] = u
    out, term = np.eye(4), np.eye(4)
    for k in range(1, 30):
        term = term @ T / k
        out = out + term
    return out[:3, :3], out[:3, 3]


def project_np(R, t, X, mdim):
    Xc = R @ X + t
    u = CAM[0] * Xc[0] / Xc[2] + CAM[2]
    v = CAM[1] * Xc[1] / Xc[2] + CAM[3]
    return np.array([u, v, u - CAM[4] / Xc[2]])[:mdim]


def test_rotate_and_matrix_agree():
    rng = np.random.default_rng(0)
    for _ in range(20):
        q, _ = rand_pose(rng)
        v = rng.normal(size=3)
        R = Rotation.from_quat(q).as_matrix()
        assert np.allclose(orc.rotate(q, v), R @ v, atol=1e-14)
        assert np.allclose(orc.quat_to_rot(q), R, atol=1e-14)
        q2 = orc.rot_to_quat(R)
        assert np.allclose(q2 * np.sign(q2[3]), q, atol=1e-13)


@pytest.mark.parametrize("mdim", [2, 3])
def test_projection(mdim):
    rng = np.random.default_rng(1)
    for _ in range(20):
        q, t = rand_pose(rng)
        R = Rotation.from_quat(q).as_matrix()
        X = R.T @ (np.array([rng.uniform(-3, 3), rng.uniform(-2, 2), rng.uniform(4, 30)]) - t)
        Xc, p = orc.project(q, t, CAM, X, mdim)
        assert np.allclose(Xc, R @ X + t, atol=1e-12)
        assert np.allclose(p, project_np(R, t, X, mdim), rtol=1e-13, atol=1e-10)


@pytest.mark.parametrize("mdim", [2, 3])
def test_jacobians_finite_differences(mdim):
    """JP = -d proj(exp(d) T X)/dd, JL = -d proj/dX (g2o sign convention)."""
    rng = np.random.default_rng(2)
    h = 1e-6
    for _ in range(10):
        q, t = rand_pose(rng)
        R = Rotation.from_quat(q).as_matrix()
        X = R.T @ (np.array([rng.uniform(-3, 3), rng.uniform(-2, 2), rng.uniform(4, 30)]) - t)
        Xc, _ = orc.project(q, t, CAM, X, mdim)
        JP, JL = orc.jacobians(Xc, q, CAM, mdim)
        numP = np.zeros((mdim, 6)); numL = np.zeros((mdim, 3))
        for k in range(6):
            d = np.zeros(6); d[k] = h
            Rp, tp = se3_exp_np(d); Rm, tm = se3_exp_np(-d)
            numP[:, k] = (project_np(Rp @ R, Rp @ t + tp, X, mdim) - project_np(Rm @ R, Rm @ t + tm, X, mdim)) / (2 * h)
        for k in range(3):
            d = np.zeros(3); d[k] = h
            numL[:, k] = (project_np(R, t, X + d, mdim) - project_np(R, t, X - d, mdim)) / (2 * h)
        assert np.allclose(JP, -numP, rtol=2e-6, atol=2e-5)
        assert np.allclose(JL, -numL, rtol=2e-6, atol=2e-5)


def test_mono_rows_equal_stereo_rows():
    rng = np.random.default_rng(3)
    q, t = rand_pose(rng)
    Xc = np.array([0.7, -0.4, 9.0])
    JP2, JL2 = orc.jacobians(Xc, q, CAM, 2)
    JP3, JL3 = orc.jacobians(Xc, q, CAM, 3)
    assert np.allclose(JP2, JP3[:2], rtol=1e-13) and np.allclose(JL2, JL3[:2], rtol=1e-13)


def test_robust_kernels():
    for kind, delta in [(0, 1.0), (1, 2.4477), (2, 3.0)]:
        for e in [1e-3, 0.5, delta**2 * 0.99, delta**2 * 1.01, 50.0, 1e4]:
            rho = orc.robustify(kind, delta, e)
            w = orc.robust_weight(kind, delta, e)
            hh = e * 1e-6
            num = (orc.robustify(kind, delta, e + hh) - orc.robustify(kind, delta, e - hh)) / (2 * hh)
            assert abs(num - w) < 1e-5 * max(1, abs(w)), (kind, e)
            if kind == 0:
                assert rho == e and w == 1
    d = 2.0
    assert orc.robustify(1, d, 100.0) == pytest.approx(2 * 10 * d - d * d)        # Huber outside
    assert orc.robustify(2, d, 100.0) == pytest.approx(d * d / 3)                  # Tukey saturates
    assert orc.robust_weight(2, d, 100.0) == 0


def test_sym3x3_inverse():
    rng = np.random.default_rng(4)
    for _ in range(20):
        A = rng.normal(size=(3, 3)); A = A @ A.T + 0.1 * np.eye(3)
        assert np.allclose(orc.sym3x3_inverse(A), np.linalg.inv(A), rtol=1e-10, atol=1e-12)


def test_se3_exp_against_mpmath():
    mp.mp.dps = 40
    rng = np.random.default_rng(5)
    for scale in [1e-7, 1e-3, 0.3, 2.0]:
        d = rng.normal(size=6) * scale
        q, t = orc.se3_exp(d)
        T = mp.zeros(4, 4)
        w = d[:3]
        T[0, 1], T[0, 2], T[1, 0], T[1, 2], T[2, 0], T[2, 1] = -w[2], w[1], w[2], -w[0], -w[1], w[0]
        for i in range(3):
            T[i, 3] = d[3 + i]
        M = mp.expm(T)
        R = np.array([[float(M[i, j]) for j in range(3)] for i in range(3)])
        tt = np.array([float(M[i, 3]) for i in range(3)])
        # the Taylor branch (theta < 1e-5) truncates at second order: error O(theta^3)
        tol = 1e-13 if scale > 1e-4 else 1e-14
        assert np.allclose(Rotation.from_quat(q).as_matrix(), R, atol=tol)
        assert np.allclose(t, tt, atol=tol * max(1, np.abs(tt).max()))


def test_pose_update_is_left_multiplication():
    rng = np.random.default_rng(6)
    for _ in range(10):
        q, t = rand_pose(rng)
        d = rng.normal(size=6) * 0.2
        q2, t2 = orc.pose_update(d, q, t)
        Re, te = se3_exp_np(d)
        R = Rotation.from_quat(q).as_matrix()
        assert np.allclose(Rotation.from_quat(q2).as_matrix(), Re @ R, atol=1e-13)
        assert np.allclose(t2, Re @ t + te, atol=1e-13)
        assert q2[3] >= 0 and abs(np.linalg.norm(q2) - 1) < 1e-14
