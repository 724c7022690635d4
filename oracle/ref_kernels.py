"""ctypes wrapper of oracle/_ref/libcuba_ref_kernels.so -- TEST INFRASTRUCTURE.

That library is the REFERENCE's own device layer (/root/reference/src/cuda_block_solver.cu, compiled in place
by oracle/ref_build/Makefile through a CUDA->HIP name shim) plus oracle/ref_build/ref_glue.cpp, which runs one
LM trial with it.  It exists only where it was built from the reference checkout (the build container); the
.so travels to the GPU box.  Used by tests/test_ref_kernels.py to pin the CPU oracle and the HIP path against
outputs of the reference itself."""
import ctypes as C
import os

import numpy as np

LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libcuba_ref_kernels.so")
_dp, _ip = C.POINTER(C.c_double), C.POINTER(C.c_int)


def available():
    return os.path.exists(LIB)


def run_trial(fp, robust, hsc_rowptr, hsc_colind, lam, xp):
    """One trial with the reference kernels. fp must list monocular edges first (flatten() does)."""
    lib = C.CDLL(LIB)
    E2, E3 = fp.E2, fp.E3
    assert np.all(fp.eDim[:E2] == 2) and np.all(fp.eDim[E2:] == 3)
    Pf, Lf, E, nblk = fp.Pf, fp.Lf, fp.E, len(hsc_colind)
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)   # noqa: E731
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)   # noqa: E731
    q, t, cam, Xw, meas, om, xp = f(fp.q), f(fp.t), f(fp.cam), f(fp.Xw), f(fp.meas), f(fp.omega), f(xp)
    eP, eL, rp, ci = i32(fp.eP), i32(fp.eL), i32(hsc_rowptr), i32(hsc_colind)
    rkt = i32([robust[0][0], robust[1][0]]); rkd = f([robust[0][1], robust[1][1]])
    out = dict(chi2=np.zeros(1), Hpp=np.zeros(36 * Pf), bp=np.zeros(6 * Pf), Hll=np.zeros(9 * Lf), bl=np.zeros(3 * Lf),
               maxdiag=np.zeros(1), bsc=np.zeros(6 * Pf), hsc=np.zeros(36 * nblk), invHll=np.zeros(9 * Lf), xl=np.zeros(3 * Lf),
               scale=np.zeros(1), q=np.zeros_like(q), t=np.zeros_like(t), Xw=np.zeros_like(Xw), chi2_after=np.zeros(1),
               chi_per_edge=np.zeros(E))
    d = lambda a: a.ctypes.data_as(_dp)   # noqa: E731
    ii = lambda a: a.ctypes.data_as(_ip)  # noqa: E731
    lib.ref_run_trial.restype = C.c_int
    rc = lib.ref_run_trial(
        C.c_int(fp.Pt), C.c_int(Pf), C.c_int(fp.Lt), C.c_int(Lf), d(q), d(t), d(cam), d(Xw),
        C.c_int(E2), C.c_int(E3), ii(eP), ii(eL), d(meas), d(om), ii(rkt), d(rkd),
        C.c_int(nblk), ii(rp), ii(ci), C.c_double(lam), d(xp),
        *[d(out[k]) for k in ("chi2", "Hpp", "bp", "Hll", "bl", "maxdiag", "bsc", "hsc", "invHll", "xl", "scale", "q", "t", "Xw",
                              "chi2_after", "chi_per_edge")])
    if rc != 0:
        raise RuntimeError(f"reference kernels reported HIP error {rc}")
    return {k: (v[0] if v.shape == (1,) else v) for k, v in out.items()}
