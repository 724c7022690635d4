"""Deterministic synthetic ORB-SLAM-style stereo BA graphs (the KITTI dataset of the reference,
samples/ba_input.7z, is not available -- see SURVEY.md section 8d).

`synth_ba(P, L, E, seed)` returns a `Graph` with exactly P poses, L landmarks and E edges:
keyframes ~1 m apart on a closed elliptical circuit driven ~1.3 times (the second lap re-observes
landmarks of the first one -> loop-closure blocks in the reduced system), landmark tracks over
consecutive keyframes, KITTI-like rectified stereo camera, ORB-pyramid noise levels, a few percent
outliers, and an initial guess perturbed on the manifold (T0 = exp(d) * T).
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation

from .graph import Graph

KITTI00_CAM = np.array([718.856, 718.856, 607.1928, 185.2157, 386.1448])
KITTI07_CAM = np.array([707.0912, 707.0912, 601.8873, 183.1104, 379.8145])
IMG_W, IMG_H = 1241.0, 376.0
Z_MIN, Z_MAX = 4.0, 40.0

# named shapes of BASELINE.json "configs"
SHAPES = {
    "kitti07": dict(P=248, L=26127, E=95037, seed=7, cam=KITTI07_CAM),
    "kitti00": dict(P=1332, L=133383, E=561116, seed=0, cam=KITTI00_CAM),
    "s2m": dict(P=5000, L=500000, E=2000000, seed=2, cam=KITTI00_CAM),
    "g4m": dict(P=10000, L=1000000, E=4000000, seed=4, cam=KITTI00_CAM),
}


def _se3_exp(d):
    """Vectorised SE3 exponential of [omega; upsilon] rows -> (R [n,3,3], t [n,3])."""
    w, u = d[:, :3], d[:, 3:]
    th = np.linalg.norm(w, axis=1)
    K = np.zeros((len(d), 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -w[:, 2], w[:, 1]
    K[:, 1, 0], K[:, 1, 2] = w[:, 2], -w[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -w[:, 1], w[:, 0]
    K2 = K @ K
    small = th < 1e-8
    ths = np.where(small, 1.0, th)
    a1 = np.where(small, 1.0, np.sin(ths) / ths)
    a2 = np.where(small, 0.5, (1 - np.cos(ths)) / ths**2)
    a3 = np.where(small, 1.0 / 6, (ths - np.sin(ths)) / ths**3)
    I = np.eye(3)[None]
    R = I + a1[:, None, None] * K + a2[:, None, None] * K2
    V = I + a2[:, None, None] * K + a3[:, None, None] * K2
    return R, np.einsum("nij,nj->ni", V, u)


def _trajectory(P, rng, lap_frac=1.3, step=1.0):
    """Camera centres and world->camera rotations for P keyframes (y axis points down)."""
    perimeter = P * step / lap_frac
    # ellipse with semi-axes a = 1.5 b; numeric arc-length parametrisation
    s = np.linspace(0, 2 * np.pi, 20001)
    ex, ez = 1.5 * np.cos(s), np.sin(s)
    seg = np.hypot(np.diff(ex), np.diff(ez))
    scale = perimeter / seg.sum()
    arc = np.concatenate([[0], np.cumsum(seg)]) * scale
    dist = np.arange(P) * step
    lap = np.floor(dist / perimeter)
    sp = np.interp(dist - lap * perimeter, arc, s)
    cx, cz = 1.5 * scale * np.cos(sp), scale * np.sin(sp)
    tx, tz = -1.5 * np.sin(sp), np.cos(sp)
    tn = np.hypot(tx, tz)
    tx, tz = tx / tn, tz / tn
    # lateral offset per lap + small jitter so laps are not identical
    off = 0.5 * lap + rng.normal(0, 0.05, P)
    cx, cz = cx + off * tz, cz - off * tx
    cy = rng.normal(0, 0.02, P)
    yaw_jit = rng.normal(0, np.deg2rad(1.0), P)
    c, s_ = np.cos(yaw_jit), np.sin(yaw_jit)
    fx_, fz_ = c * tx + s_ * tz, -s_ * tx + c * tz          # forward (camera z) in world
    Rwc = np.zeros((P, 3, 3))
    Rwc[:, :, 2] = np.stack([fx_, np.zeros(P), fz_], 1)      # z_cam
    Rwc[:, :, 1] = np.array([0.0, 1.0, 0.0])                 # y_cam (down)
    Rwc[:, :, 0] = np.cross(Rwc[:, :, 1], Rwc[:, :, 2])      # x_cam = y x z
    Rcw = np.transpose(Rwc, (0, 2, 1))
    C = np.stack([cx, cy, cz], 1)
    t = -np.einsum("nij,nj->ni", Rcw, C)
    lap_len = int(round(perimeter / step))
    return Rcw, t, lap_len


def _project(Rcw, t, X, cam):
    Xc = np.einsum("...ij,...j->...i", Rcw, X) + t
    z = Xc[..., 2]
    zs = np.where(np.abs(z) < 1e-9, 1e-9, z)
    u = cam[0] * Xc[..., 0] / zs + cam[2]
    v = cam[1] * Xc[..., 1] / zs + cam[3]
    ok = (z >= Z_MIN) & (z <= Z_MAX + 20) & (u >= 0) & (u < IMG_W) & (v >= 0) & (v < IMG_H)
    return u, v, z, ok


def synth_ba(P, L, E, seed=0, cam=KITTI00_CAM, stereo_frac=0.85, outlier_frac=0.03,
             max_track=40, loop_closure=True, perturb=True, fix_first=True) -> Graph:
    if E < 2 * L:
        raise ValueError("need at least two observations per landmark (E >= 2 L)")
    rng = np.random.default_rng(seed)
    cam = np.asarray(cam, dtype=np.float64)
    Rcw, t, lap_len = _trajectory(P, rng)
    mean_track = E / L
    NW = 6 if (loop_closure and lap_len + 8 < P) else 0       # loop-closure window (frames of the other lap)

    cand_obs_p, cand_obs_l, cand_X, cand_n = [], [], [], []
    n_have, e_have = 0, 0
    # geometric track-length model: n = 2 + Geom; visibility filtering removes some, so aim higher
    p_geo = min(0.95, 1.0 / max(mean_track * 1.35 - 1.0, 1.05))

    def gen_batch():
        nonlocal n_have, e_have
        N = int(1.2 * L) + 256
        a = rng.integers(0, max(1, P - int(mean_track)), N) if P < 64 else rng.integers(0, P, N)
        n = np.minimum(2 + rng.geometric(p_geo, N) - 1, max_track)
        # point defined in the LAST frame of its track so that it stays in front of the camera
        last = np.minimum(a + n - 1, P - 1)
        zl = rng.uniform(Z_MIN + 0.5, Z_MAX - 1.0, N)
        ul = rng.uniform(40, IMG_W - 40, N)
        vl = rng.uniform(20, IMG_H - 20, N)
        Xc = np.stack([(ul - cam[2]) / cam[0] * zl, (vl - cam[3]) / cam[1] * zl, zl], 1)
        Xw = np.einsum("nji,nj->ni", Rcw[last], Xc - t[last])
        K = max_track + NW
        frames = a[:, None] + np.arange(max_track)[None, :]
        in_track = (np.arange(max_track)[None, :] < n[:, None]) & (frames < P)
        if NW:
            partner = np.where(a + lap_len < P, a + lap_len, a - lap_len)
            lc = partner[:, None] + np.arange(-2, NW - 2)[None, :]
            lc_ok = (partner[:, None] >= 0) & (lc >= 0) & (lc < P) & (rng.random((N, NW)) < 0.6)
            frames = np.concatenate([frames, lc], 1)
            in_track = np.concatenate([in_track, lc_ok], 1)
        fr = np.clip(frames, 0, P - 1)
        ok = np.zeros((N, K), dtype=bool)
        for k0 in range(0, K, 8):                               # chunk over the frame axis to bound memory
            sl = slice(k0, min(k0 + 8, K))
            _, _, _, vis = _project(Rcw[fr[:, sl]], t[fr[:, sl]], Xw[:, None, :], cam)
            ok[:, sl] = vis & in_track[:, sl]
        cnt = ok.sum(1)
        good = cnt >= 2
        li, ki = np.nonzero(ok & good[:, None])
        new_index = np.cumsum(good) - 1 + n_have
        cand_obs_p.append(fr[li, ki]); cand_obs_l.append(new_index[li])
        cand_X.append(Xw[good]); cand_n.append(cnt[good])
        n_have += int(good.sum()); e_have += int(cnt[good].sum())

    def select(n_all):
        """Choose exactly L landmarks whose tracks hold at least E observations (None if impossible yet)."""
        chosen = np.zeros(len(n_all), dtype=bool)
        chosen[:L] = True
        S = int(n_all[:L].sum())
        if S < E:
            unused = np.nonzero(~chosen)[0]
            unused = unused[np.argsort(-n_all[unused], kind="stable")]
            used = np.nonzero(chosen)[0]
            used = used[np.argsort(n_all[used], kind="stable")]
            m = min(len(unused), len(used))
            gain = np.cumsum(n_all[unused[:m]] - n_all[used[:m]])
            k = int(np.searchsorted(gain, E - S)) + 1
            if k > m or gain[k - 1] < E - S:
                return None
            chosen[used[:k]] = False
            chosen[unused[:k]] = True
        return chosen

    chosen = None
    for _ in range(60):
        gen_batch()
        if n_have < int(1.6 * L) + 64 or e_have < int(1.25 * E) + 64:
            continue
        n_all = np.concatenate(cand_n)
        chosen = select(n_all)
        if chosen is not None:
            break
    if chosen is None:
        raise RuntimeError("could not generate enough visible landmark tracks for the requested (P, L, E)")
    obs_p = np.concatenate(cand_obs_p); obs_l = np.concatenate(cand_obs_l)
    X_all = np.concatenate(cand_X)
    S = int(n_all[chosen].sum())
    keep_n = n_all.copy()
    surplus = S - E
    if surplus > 0:
        idx = np.nonzero(chosen)[0]
        spare = np.repeat(idx, np.maximum(keep_n[idx] - 2, 0))
        if len(spare) < surplus:
            raise RuntimeError("cannot trim tracks down to the requested edge count")
        drop = rng.choice(len(spare), size=surplus, replace=False)
        np.subtract.at(keep_n, spare[drop], 1)
    # observation rank within its landmark (obs are grouped by landmark, ascending)
    first = np.concatenate([[0], np.cumsum(n_all)[:-1]])
    rank = np.arange(len(obs_l)) - first[obs_l]
    sel = chosen[obs_l] & (rank < keep_n[obs_l])
    new_l = np.cumsum(chosen) - 1
    ep = obs_p[sel].astype(np.int64)
    el = new_l[obs_l[sel]].astype(np.int64)
    Xw_true = X_all[chosen]
    assert len(Xw_true) == L and len(ep) == E, (len(Xw_true), L, len(ep), E)

    # ---- measurements ----------------------------------------------------------------------
    u, v, z, _ = _project(Rcw[ep], t[ep], Xw_true[el], cam)
    lvl = rng.integers(0, 8, E)
    sigma = 1.2 ** lvl
    info = 1.2 ** (-2.0 * lvl)
    noise = rng.normal(0, 1, (E, 3)) * sigma[:, None]
    outl = rng.random(E) < outlier_frac
    noise[outl] += rng.uniform(-30, 30, (int(outl.sum()), 3))
    meas = np.stack([u, v, u - cam[4] / z], 1) + noise
    is_stereo = rng.random(E) < stereo_frac

    # ---- initial guess ------------------------------------------------------------------------
    q_true = Rotation.from_matrix(Rcw).as_quat()               # (x,y,z,w)
    q_true[q_true[:, 3] < 0] *= -1
    if perturb:
        d = np.concatenate([rng.normal(0, np.deg2rad(0.5), (P, 3)), rng.normal(0, 0.05, (P, 3))], 1)
        if fix_first:
            d[0] = 0
        dR, dt = _se3_exp(d)
        R0 = dR @ Rcw
        t0 = np.einsum("nij,nj->ni", dR, t) + dt
        X0 = Xw_true + rng.normal(0, 0.10, (L, 3))
    else:
        R0, t0, X0 = Rcw, t.copy(), Xw_true.copy()
    q0 = Rotation.from_matrix(R0).as_quat()
    q0[q0[:, 3] < 0] *= -1

    pose_fixed = np.zeros(P, dtype=bool)
    if fix_first:
        pose_fixed[0] = True
    m_idx, s_idx = np.nonzero(~is_stereo)[0], np.nonzero(is_stereo)[0]
    return Graph(
        pose_ids=np.arange(P, dtype=np.int64), pose_fixed=pose_fixed, pose_q=q0, pose_t=t0,
        pose_cam=np.tile(cam, (P, 1)),
        lm_ids=np.arange(P, P + L, dtype=np.int64), lm_fixed=np.zeros(L, dtype=bool), lm_X=X0,
        mono_vp=ep[m_idx], mono_vl=el[m_idx] + P, mono_meas=meas[m_idx, :2].copy(), mono_info=info[m_idx],
        stereo_vp=ep[s_idx], stereo_vl=el[s_idx] + P, stereo_meas=meas[s_idx].copy(), stereo_info=info[s_idx],
        truth=dict(q=q_true, t=t.copy(), Xw=Xw_true),
    )


def synth_named(name: str, **over) -> Graph:
    """One of the BASELINE.json shapes: kitti07, kitti00, s2m, g4m."""
    kw = dict(SHAPES[name]); kw.update(over)
    return synth_ba(**kw)
