"""Graph containers and the host-side "initialize" step (graph -> solver-order flat arrays).

`Graph` is the user-level view (vertex ids, fixed flags, mono / stereo edge lists) -- the same
information the reference keeps in `CudaBundleAdjustmentImpl` (src/cuda_bundle_adjustment.cpp:677-903,
containers :893-902).  `flatten()` restates `CudaBlockSolver::initialize`
(src/cuda_bundle_adjustment.cpp:115-261): vertices are visited in id order, vertices without edges
are skipped, free vertices are indexed before fixed ones, edges with both ends fixed are dropped,
monocular edges come before stereo edges.  Unlike the reference (unordered_set iteration order) the
edge order inside each type is the insertion order, so results are reproducible.

JSON I/O follows the schema read by samples/sample_ba_from_file.cpp:91-157.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field

import numpy as np

ROBUST_NONE, ROBUST_HUBER, ROBUST_TUKEY = 0, 1, 2   # include/cuda_bundle_adjustment_types.h:213-218
EDGE_MONO, EDGE_STEREO = 0, 1                       # include/cuda_bundle_adjustment_types.h:143-148


@dataclass
class Graph:
    """User-level bundle-adjustment graph (one camera model per pose, like `PoseVertex::camera`)."""
    pose_ids: np.ndarray        # [P] int
    pose_fixed: np.ndarray      # [P] bool
    pose_q: np.ndarray          # [P,4] (x,y,z,w), world->camera
    pose_t: np.ndarray          # [P,3]
    pose_cam: np.ndarray        # [P,5] fx fy cx cy bf
    lm_ids: np.ndarray          # [L] int
    lm_fixed: np.ndarray        # [L] bool
    lm_X: np.ndarray            # [L,3]
    mono_vp: np.ndarray         # [E2] pose ids
    mono_vl: np.ndarray         # [E2] landmark ids
    mono_meas: np.ndarray       # [E2,2]
    mono_info: np.ndarray       # [E2]
    stereo_vp: np.ndarray       # [E3]
    stereo_vl: np.ndarray       # [E3]
    stereo_meas: np.ndarray     # [E3,3]
    stereo_info: np.ndarray     # [E3]
    truth: dict = field(default_factory=dict)   # optional ground truth (synthetic graphs)

    @property
    def nposes(self): return len(self.pose_ids)

    @property
    def nlandmarks(self): return len(self.lm_ids)

    @property
    def nedges(self): return len(self.mono_vp) + len(self.stereo_vp)

    def to_json(self, path):
        cam = self.pose_cam[0]
        doc = {
            "fx": float(cam[0]), "fy": float(cam[1]), "cx": float(cam[2]), "cy": float(cam[3]), "bf": float(cam[4]),
            "pose_vertices": [
                {"id": int(i), "fixed": int(f), "q": [float(v) for v in q], "t": [float(v) for v in t]}
                for i, f, q, t in zip(self.pose_ids, self.pose_fixed, self.pose_q, self.pose_t)],
            "landmark_vertices": [
                {"id": int(i), "fixed": int(f), "Xw": [float(v) for v in X]}
                for i, f, X in zip(self.lm_ids, self.lm_fixed, self.lm_X)],
            "monocular_edges": [
                {"vertexP": int(p), "vertexL": int(l), "measurement": [float(v) for v in m], "information": float(w)}
                for p, l, m, w in zip(self.mono_vp, self.mono_vl, self.mono_meas, self.mono_info)],
            "stereo_edges": [
                {"vertexP": int(p), "vertexL": int(l), "measurement": [float(v) for v in m], "information": float(w)}
                for p, l, m, w in zip(self.stereo_vp, self.stereo_vl, self.stereo_meas, self.stereo_info)],
        }
        with open(path, "w") as f:
            json.dump(doc, f)

    @staticmethod
    def from_json(path) -> "Graph":
        with open(path) as f:
            text = f.read()
        if text.startswith("%YAML"):           # OpenCV FileStorage may prepend a YAML directive
            text = text.split("\n", 1)[1]
        d = json.loads(text)
        cam = np.array([d["fx"], d["fy"], d["cx"], d["cy"], d["bf"]], dtype=np.float64)
        pv, lv = d["pose_vertices"], d["landmark_vertices"]
        me, se = d.get("monocular_edges", []), d.get("stereo_edges", [])
        f64 = np.float64
        return Graph(
            pose_ids=np.array([v["id"] for v in pv], dtype=np.int64),
            pose_fixed=np.array([bool(v["fixed"]) for v in pv], dtype=bool),
            pose_q=np.array([v["q"] for v in pv], dtype=f64).reshape(-1, 4),
            pose_t=np.array([v["t"] for v in pv], dtype=f64).reshape(-1, 3),
            pose_cam=np.tile(cam, (len(pv), 1)),
            lm_ids=np.array([v["id"] for v in lv], dtype=np.int64),
            lm_fixed=np.array([bool(v["fixed"]) for v in lv], dtype=bool),
            lm_X=np.array([v["Xw"] for v in lv], dtype=f64).reshape(-1, 3),
            mono_vp=np.array([e["vertexP"] for e in me], dtype=np.int64),
            mono_vl=np.array([e["vertexL"] for e in me], dtype=np.int64),
            mono_meas=np.array([e["measurement"] for e in me], dtype=f64).reshape(-1, 2),
            mono_info=np.array([e["information"] for e in me], dtype=f64),
            stereo_vp=np.array([e["vertexP"] for e in se], dtype=np.int64),
            stereo_vl=np.array([e["vertexL"] for e in se], dtype=np.int64),
            stereo_meas=np.array([e["measurement"] for e in se], dtype=f64).reshape(-1, 3),
            stereo_info=np.array([e["information"] for e in se], dtype=f64),
        )


@dataclass
class FlatProblem:
    """Solver-order arrays: exactly what crosses the C ABI (`cuba_hip_set_graph`)."""
    Pt: int
    Pf: int
    Lt: int
    Lf: int
    q: np.ndarray       # [Pt,4]
    t: np.ndarray       # [Pt,3]
    cam: np.ndarray     # [Pt,5]
    Xw: np.ndarray      # [Lt,3]
    eP: np.ndarray      # [E] int32 solver pose index
    eL: np.ndarray      # [E] int32 solver landmark index
    eDim: np.ndarray    # [E] uint8, 2 = mono, 3 = stereo
    meas: np.ndarray    # [E,3] (third component 0 for mono)
    omega: np.ndarray   # [E]
    pose_src: np.ndarray  # [Pt] row of Graph.pose_* each solver pose came from
    lm_src: np.ndarray    # [Lt]
    edge_src: np.ndarray  # [E] index into concat(mono, stereo) edge lists of the Graph

    @property
    def E(self): return len(self.eP)

    @property
    def E2(self): return int((self.eDim == 2).sum())

    @property
    def E3(self): return int((self.eDim == 3).sum())


def flatten(g: Graph) -> FlatProblem:
    """Restates CudaBlockSolver::initialize (src/cuda_bundle_adjustment.cpp:115-261)."""
    P, L = g.nposes, g.nlandmarks
    pid_sorted = np.argsort(g.pose_ids, kind="stable")
    lid_sorted = np.argsort(g.lm_ids, kind="stable")
    # id -> row lookup
    prow = {int(i): r for r, i in enumerate(g.pose_ids)}
    lrow = {int(i): r for r, i in enumerate(g.lm_ids)}
    vp = np.concatenate([g.mono_vp, g.stereo_vp]).astype(np.int64)
    vl = np.concatenate([g.mono_vl, g.stereo_vl]).astype(np.int64)
    if len(vp):
        if P and int(g.pose_ids.min()) >= 0 and int(g.pose_ids.max()) < 4 * P + 1024:
            lut = np.full(int(g.pose_ids.max()) + 1, -1, dtype=np.int64)
            lut[g.pose_ids] = np.arange(P)
            if vp.min() < 0 or vp.max() >= len(lut):
                raise KeyError("edge references an unknown pose id")
            ep_row = lut[vp]
        else:
            ep_row = np.array([prow[int(i)] for i in vp], dtype=np.int64)
        if L and int(g.lm_ids.min()) >= 0 and int(g.lm_ids.max()) < 4 * (L + P) + 1024:
            lut = np.full(int(g.lm_ids.max()) + 1, -1, dtype=np.int64)
            lut[g.lm_ids] = np.arange(L)
            if vl.min() < 0 or vl.max() >= len(lut):
                raise KeyError("edge references an unknown landmark id")
            el_row = lut[vl]
        else:
            el_row = np.array([lrow[int(i)] for i in vl], dtype=np.int64)
        if (ep_row < 0).any() or (el_row < 0).any():
            raise KeyError("edge references an unknown vertex id")   # map::at throws in the reference (:707-715)
    else:
        ep_row = np.zeros(0, dtype=np.int64)
        el_row = np.zeros(0, dtype=np.int64)
    has_edge_p = np.zeros(P, dtype=bool); has_edge_p[ep_row] = True
    has_edge_l = np.zeros(L, dtype=bool); has_edge_l[el_row] = True

    def order(sorted_rows, has_edge, fixed):
        rows = [r for r in sorted_rows if has_edge[r]]
        free = [r for r in rows if not fixed[r]]
        fix = [r for r in rows if fixed[r]]
        return np.array(free + fix, dtype=np.int64), len(free)

    pose_src, Pf = order(pid_sorted, has_edge_p, g.pose_fixed)
    lm_src, Lf = order(lid_sorted, has_edge_l, g.lm_fixed)
    iP = np.full(P, -1, dtype=np.int64); iP[pose_src] = np.arange(len(pose_src))
    iL = np.full(L, -1, dtype=np.int64); iL[lm_src] = np.arange(len(lm_src))

    E2 = len(g.mono_vp)
    dim = np.concatenate([np.full(E2, 2, np.uint8), np.full(len(g.stereo_vp), 3, np.uint8)])
    meas = np.zeros((len(vp), 3), dtype=np.float64)
    meas[:E2, :2] = g.mono_meas
    meas[E2:, :] = g.stereo_meas
    info = np.concatenate([g.mono_info, g.stereo_info]).astype(np.float64)
    active = ~(g.pose_fixed[ep_row] & g.lm_fixed[el_row]) if len(vp) else np.zeros(0, dtype=bool)
    keep = np.nonzero(active)[0]
    return FlatProblem(
        Pt=len(pose_src), Pf=Pf, Lt=len(lm_src), Lf=Lf,
        q=np.ascontiguousarray(g.pose_q[pose_src], dtype=np.float64),
        t=np.ascontiguousarray(g.pose_t[pose_src], dtype=np.float64),
        cam=np.ascontiguousarray(g.pose_cam[pose_src], dtype=np.float64),
        Xw=np.ascontiguousarray(g.lm_X[lm_src], dtype=np.float64),
        eP=iP[ep_row[keep]].astype(np.int32), eL=iL[el_row[keep]].astype(np.int32),
        eDim=np.ascontiguousarray(dim[keep]), meas=np.ascontiguousarray(meas[keep]),
        omega=np.ascontiguousarray(info[keep]),
        pose_src=pose_src, lm_src=lm_src, edge_src=keep,
    )


def write_back(g: Graph, fp: FlatProblem, q, t, Xw):
    """`finalize()`: copy optimised estimates back into the user graph (src/cuda_bundle_adjustment.cpp:512-526)."""
    g.pose_q[fp.pose_src] = np.asarray(q).reshape(-1, 4)
    g.pose_t[fp.pose_src] = np.asarray(t).reshape(-1, 3)
    g.lm_X[fp.lm_src] = np.asarray(Xw).reshape(-1, 3)
