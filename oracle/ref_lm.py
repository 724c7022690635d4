"""ctypes wrapper of oracle/_ref/libcuba_ref_lm.so -- TEST INFRASTRUCTURE.

That library is the REFERENCE's whole optimiser compiled from its own sources in place (oracle/ref_build/Makefile):
src/cuda_bundle_adjustment.cpp (CudaBundleAdjustmentImpl::optimize + CudaBlockSolver), src/sparse_block_matrix.cpp and
src/cuda_block_solver.cu through the CUDA->HIP name shim, with the closed-source cuSOLVER step replaced by an exact host
Cholesky (ref_linear_solver.cpp).  Used by tests/test_ref_lm.py to pin LM TRAJECTORIES of the CPU oracle and of the HIP
path against the reference itself.  It exists only where it was built from the reference checkout; the .so travels to
the GPU box."""
import ctypes as C
import os

import numpy as np

LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libcuba_ref_lm.so")
_dp, _ip, _bp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_uint8)


# the reference's CudaBundleAdjustment::timeProfile() keys (src/cuda_bundle_adjustment.cpp:545-562), seconds of the last initialize() + optimize()
PROFILE_KEYS = ("0: Initialize Optimizer", "1: Build Structure", "2: Compute Error", "3: Build System", "4: Schur Complement",
                "5: Symbolic Decomposition", "6: Numerical Decomposition", "7: Update Solution")


def available():
    return os.path.exists(LIB)


def run(g, robust, niterations, nruns=1):
    """initialize() + optimize(niterations), `nruns` times in a row, on a cuba_amd.graph.Graph (user-level ids).
    Returns dict(chi2 of the LAST run, q, t, Xw in the graph's row order, per-edge chi2 mono / stereo, and the reference's own stage
    timers + the host wall of the last initialize() / optimize())."""
    lib = C.CDLL(LIB)
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)   # noqa: E731
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)   # noqa: E731
    u8 = lambda a: np.ascontiguousarray(a, dtype=np.uint8)    # noqa: E731
    d = lambda a: a.ctypes.data_as(_dp)   # noqa: E731
    ii = lambda a: a.ctypes.data_as(_ip)  # noqa: E731
    P, L, E2, E3 = g.nposes, g.nlandmarks, len(g.mono_vp), len(g.stereo_vp)
    pid, pfix, q, t, cam = i32(g.pose_ids), u8(g.pose_fixed), f(g.pose_q), f(g.pose_t), f(g.pose_cam)
    lid, lfix, X = i32(g.lm_ids), u8(g.lm_fixed), f(g.lm_X)
    mvp, mvl, mm, mi = i32(g.mono_vp), i32(g.mono_vl), f(g.mono_meas), f(g.mono_info)
    svp, svl, sm, si = i32(g.stereo_vp), i32(g.stereo_vl), f(g.stereo_meas), f(g.stereo_info)
    rkt = i32([robust[0][0], robust[1][0]]); rkd = f([robust[0][1], robust[1][1]])
    chi2 = np.zeros(max(niterations, 1)); n = C.c_int()
    qo, to, Xo = np.zeros_like(q), np.zeros_like(t), np.zeros_like(X)
    cm, cs = np.zeros(max(E2, 1)), np.zeros(max(E3, 1))
    lib.ref_lm_run.restype = C.c_int
    rc = lib.ref_lm_run(C.c_int(P), ii(pid), pfix.ctypes.data_as(_bp), d(q), d(t), d(cam),
                        C.c_int(L), ii(lid), lfix.ctypes.data_as(_bp), d(X),
                        C.c_int(E2), ii(mvp), ii(mvl), d(mm), d(mi), C.c_int(E3), ii(svp), ii(svl), d(sm), d(si),
                        ii(rkt), d(rkd), C.c_int(niterations), C.c_int(nruns),
                        d(chi2), C.byref(n), d(qo), d(to), d(Xo), d(cm), d(cs))
    if rc != 0:
        raise RuntimeError(f"reference optimiser returned {rc}")
    prof = np.zeros(10)
    lib.ref_lm_last_profile(d(prof))
    return dict(chi2=chi2[:n.value], q=qo, t=to, Xw=Xo, chi_mono=cm[:E2], chi_stereo=cs[:E3],
                profile=dict(zip(PROFILE_KEYS, prof[:8].tolist())), wall_initialize=float(prof[8]), wall_optimize=float(prof[9]))
