"""Host-side logic that runs without a GPU: graph flattening (the reference's initialize()), JSON I/O,
the synthetic generator, and the C-ABI library's exported surface."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, with_fixed
from cuba_amd.graph import Graph, flatten, write_back
from cuba_amd.synth import SHAPES, synth_ba


def tiny_graph():
    cam = np.array([500.0, 500.0, 320.0, 240.0, 50.0])
    return Graph(
        pose_ids=np.array([7, 3, 5, 9]), pose_fixed=np.array([False, True, False, False]),
        pose_q=np.tile([0, 0, 0, 1.0], (4, 1)), pose_t=np.arange(12, dtype=float).reshape(4, 3), pose_cam=np.tile(cam, (4, 1)),
        lm_ids=np.array([20, 10, 30, 40]), lm_fixed=np.array([False, True, False, False]),
        lm_X=np.arange(12, dtype=float).reshape(4, 3) + 100,
        mono_vp=np.array([7, 3]), mono_vl=np.array([20, 10]), mono_meas=np.array([[1.0, 2.0], [3.0, 4.0]]), mono_info=np.array([1.0, 2.0]),
        stereo_vp=np.array([3, 5, 7]), stereo_vl=np.array([20, 30, 30]),
        stereo_meas=np.array([[1.0, 2, 3], [4, 5, 6], [7, 8, 9]]), stereo_info=np.array([3.0, 4.0, 5.0]),
    )


def test_flatten_follows_reference_initialize():
    g = tiny_graph()
    fp = flatten(g)
    # pose 9 and landmark 40 have no edges -> skipped (src/cuda_bundle_adjustment.cpp:144,165)
    # id order, free first: poses 5,7 then fixed 3 ; landmarks 20,30 then fixed 10
    assert list(g.pose_ids[fp.pose_src]) == [5, 7, 3] and fp.Pf == 2 and fp.Pt == 3
    assert list(g.lm_ids[fp.lm_src]) == [20, 30, 10] and fp.Lf == 2 and fp.Lt == 3
    # the (pose 3 fixed, landmark 10 fixed) mono edge is dropped (:209-221); mono edges precede stereo edges
    assert fp.E == 4 and list(fp.eDim) == [2, 3, 3, 3]
    assert list(fp.eP) == [1, 2, 0, 1] and list(fp.eL) == [0, 0, 1, 1]
    assert list(fp.edge_src) == [0, 2, 3, 4]
    assert np.array_equal(fp.meas[0], [1.0, 2.0, 0.0]) and np.array_equal(fp.meas[3], [7.0, 8.0, 9.0])
    assert list(fp.omega) == [1.0, 3.0, 4.0, 5.0]
    # finalize(): results land in the user's rows (:512-526)
    q = np.tile([0, 0, 1.0, 0], (3, 1)); t = np.full((3, 3), -1.0); X = np.full((3, 3), -2.0)
    write_back(g, fp, q, t, X)
    assert np.all(g.pose_t[[0, 1, 2]] == -1) and np.array_equal(g.pose_t[3], [9, 10, 11.0])
    assert np.all(g.lm_X[[0, 1, 2]] == -2) and np.array_equal(g.lm_X[3], [109, 110, 111.0])


def test_flatten_unknown_vertex_raises():
    g = tiny_graph()
    g.mono_vp[0] = 12345
    with pytest.raises(KeyError):
        flatten(g)


def test_json_roundtrip(tmp_path):
    g = synth_ba(12, 150, 600, seed=2)
    path = tmp_path / "g.json"
    g.to_json(path)
    h = Graph.from_json(path)
    for name in ("pose_ids", "pose_fixed", "pose_q", "pose_t", "lm_ids", "lm_X", "mono_vp", "mono_vl", "mono_meas",
                 "mono_info", "stereo_vp", "stereo_vl", "stereo_meas", "stereo_info"):
        assert np.array_equal(getattr(g, name), getattr(h, name)), name
    a, b = flatten(g), flatten(h)
    assert np.array_equal(a.eP, b.eP) and np.array_equal(a.meas, b.meas)


def test_synth_exact_counts_and_determinism():
    g1 = synth_ba(40, 600, 2400, seed=1)
    g2 = synth_ba(40, 600, 2400, seed=1)
    assert (g1.nposes, g1.nlandmarks, g1.nedges) == (40, 600, 2400)
    assert np.array_equal(g1.stereo_meas, g2.stereo_meas) and np.array_equal(g1.pose_q, g2.pose_q)
    fp = flatten(g1)
    n = np.bincount(fp.eL, minlength=fp.Lt)
    assert n.min() >= 2                       # every landmark is observed at least twice
    assert fp.Pf == 39 and fp.Pt == 40       # pose 0 fixes the gauge
    assert abs(np.linalg.norm(g1.pose_q, axis=1) - 1).max() < 1e-12
    # all observations are in front of the camera and inside the image at the ground truth
    assert set(SHAPES) >= {"kitti07", "kitti00", "s2m", "g4m"}


def test_capi_library_exports_every_declared_symbol():
    """The C-ABI .so loads on a CPU-only box and exports exactly what include/cuba_hip.h declares."""
    from cuba_amd import capi
    header = open(os.path.join(ROOT, "include", "cuba_hip.h")).read()
    declared = sorted(set(re.findall(r"^(?:int|const char\*)\s+(cuba_hip_[a-z_0-9]+)\s*\(", header, re.M)))
    assert len(declared) >= 30
    capi.build_library()
    for path in (capi.LIB_PATH, capi.LIB_PATH_F32):          # fp64 build and the USE_FLOAT32-style fp32 build
        lib = ctypes.CDLL(path)
        for name in declared:
            assert hasattr(lib, name), f"{name} declared in cuba_hip.h but not exported by {path}"
    assert capi.load_library("f64").cuba_hip_scalar_size() == 8
    assert capi.load_library("f32").cuba_hip_scalar_size() == 4


def test_capi_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cuba_amd.capi import CubaHipError, HipSolver
    with pytest.raises(CubaHipError):
        HipSolver()
