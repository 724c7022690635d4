"""ctypes wrapper of oracle/libba_oracle.so -- TEST INFRASTRUCTURE (see ba_oracle.cpp header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libba_oracle.so")
_LIB_OMP = os.path.join(_HERE, "libba_oracle_omp.so")     # same source with OpenMP edge/landmark loops: CPU-baseline timing only
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force=False, omp=False):
    src = os.path.join(_HERE, "ba_oracle.cpp")
    lib = _LIB_OMP if omp else _LIB
    if force or not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", os.path.basename(lib)])
    return lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _load(omp=False):
    lib = C.CDLL(build(omp=omp))
    lib.orc_max_threads.restype = C.c_int
    lib.orc_set_threads.restype = C.c_int
    lib.orc_set_threads.argtypes = [C.c_int]
    lib.orc_create.restype = C.c_void_p
    lib.orc_create.argtypes = [C.c_int] * 4 + [_dp] * 4 + [C.c_int, _ip, _ip, C.POINTER(C.c_uint8), _dp, _dp]
    for name in ("orc_compute_errors", "orc_max_diagonal"):
        getattr(lib, name).restype = C.c_double
        getattr(lib, name).argtypes = [C.c_void_p]
    lib.orc_compute_scale.restype = C.c_double
    lib.orc_compute_scale.argtypes = [C.c_void_p, C.c_double]
    lib.orc_set_lambda.argtypes = [C.c_void_p, C.c_double]
    lib.orc_set_robust_kernel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
    for name in ("orc_destroy", "orc_build_structure", "orc_build_system", "orc_restore_diagonal", "orc_schur",
                 "orc_update", "orc_push", "orc_pop"):
        getattr(lib, name).argtypes = [C.c_void_p]
        getattr(lib, name).restype = None
    lib.orc_solve.argtypes = [C.c_void_p]
    lib.orc_solve.restype = C.c_int
    lib.orc_solve_reduced.argtypes = [C.c_void_p]
    lib.orc_solve_reduced.restype = C.c_int
    lib.orc_back_substitute.argtypes = [C.c_void_p]
    lib.orc_back_substitute.restype = None
    lib.orc_set_array.argtypes = [C.c_void_p, C.c_int, _dp]
    lib.orc_set_array.restype = C.c_long
    lib.orc_optimize.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _ip]
    lib.orc_optimize.restype = C.c_int
    lib.orc_chi_squares.argtypes = [C.c_void_p, _dp]
    lib.orc_get_state.argtypes = [C.c_void_p, _dp, _dp, _dp]
    lib.orc_set_state.argtypes = [C.c_void_p, _dp, _dp, _dp]
    lib.orc_hsc_nblocks.argtypes = [C.c_void_p]
    lib.orc_hsc_nblocks.restype = C.c_int
    lib.orc_get_hsc.argtypes = [C.c_void_p, _ip, _ip, _dp]
    lib.orc_get_array.argtypes = [C.c_void_p, C.c_int, _dp]
    lib.orc_get_array.restype = C.c_long
    lib.orc_robustify.restype = C.c_double
    lib.orc_robustify.argtypes = [C.c_int, C.c_double, C.c_double]
    lib.orc_robust_weight.restype = C.c_double
    lib.orc_robust_weight.argtypes = [C.c_int, C.c_double, C.c_double]
    lib.orc_block_cholesky_solve.restype = C.c_int
    lib.orc_block_cholesky_solve.argtypes = [C.c_int, _ip, _ip, _dp, _dp, _dp]
    return lib


_lib = None
_lib_omp = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def lib_omp():
    """OpenMP build (timing baseline only; parity checks use lib())."""
    global _lib_omp
    if _lib_omp is None:
        _lib_omp = _load(omp=True)
    return _lib_omp


# ---- unit-level helpers -------------------------------------------------------------------------
def project(q, t, cam, Xw, mdim):
    q, t, cam, Xw = (np.ascontiguousarray(a, dtype=np.float64) for a in (q, t, cam, Xw))
    Xc, p = np.zeros(3), np.zeros(3)
    lib().orc_project(_d(q), _d(t), _d(cam), _d(Xw), C.c_int(mdim), _d(Xc), _d(p))
    return Xc, p[:mdim]


def jacobians(Xc, q, cam, mdim):
    Xc, q, cam = (np.ascontiguousarray(a, dtype=np.float64) for a in (Xc, q, cam))
    JP, JL = np.zeros(mdim * 6), np.zeros(mdim * 3)
    lib().orc_jacobians(_d(Xc), _d(q), _d(cam), C.c_int(mdim), _d(JP), _d(JL))
    return JP.reshape(6, mdim).T.copy(), JL.reshape(3, mdim).T.copy()   # column-major -> [mdim,6], [mdim,3]


def robustify(kind, delta, e):
    return lib().orc_robustify(kind, delta, e)


def robust_weight(kind, delta, e):
    return lib().orc_robust_weight(kind, delta, e)


def sym3x3_inverse(A):
    A = np.asfortranarray(A, dtype=np.float64)
    B = np.zeros((3, 3), order="F")
    lib().orc_sym3x3_inverse(_d(A), _d(B))
    return np.array(B)


def rotate(q, v):
    q, v = np.ascontiguousarray(q, dtype=np.float64), np.ascontiguousarray(v, dtype=np.float64)
    o = np.zeros(3)
    lib().orc_rotate(_d(q), _d(v), _d(o))
    return o


def quat_to_rot(q):
    q = np.ascontiguousarray(q, dtype=np.float64)
    R = np.zeros((3, 3), order="F")
    lib().orc_quat_to_rot(_d(q), _d(R))
    return np.array(R)


def rot_to_quat(R):
    R = np.asfortranarray(R, dtype=np.float64)
    q = np.zeros(4)
    lib().orc_rot_to_quat(_d(R), _d(q))
    return q


def se3_exp(upd):
    upd = np.ascontiguousarray(upd, dtype=np.float64)
    q, t = np.zeros(4), np.zeros(3)
    lib().orc_se3_exp(_d(upd), _d(q), _d(t))
    return q, t


def pose_update(upd, q, t):
    upd = np.ascontiguousarray(upd, dtype=np.float64)
    q, t = np.array(q, dtype=np.float64), np.array(t, dtype=np.float64)
    lib().orc_pose_update(_d(upd), _d(q), _d(t))
    return q, t


def block_cholesky_solve(rowptr, colind, values, b):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    colind = np.ascontiguousarray(colind, dtype=np.int32)
    values = np.ascontiguousarray(values, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros_like(b)
    rc = lib().orc_block_cholesky_solve(len(rowptr) - 1, rowptr.ctypes.data_as(_ip), colind.ctypes.data_as(_ip),
                                        _d(values), _d(b), _d(x))
    return rc, x


# ---- problem-level ------------------------------------------------------------------------------
class OracleSolver:
    """Mirror of CudaBlockSolver's stage methods on the CPU oracle; takes a FlatProblem."""
    ARR = dict(Hpp=0, bp=1, Hll=2, bl=3, Hpl=4, bsc=5, xp=6, xl=7, invHll=8, err=9, Xc=10)

    def __init__(self, fp, robust=((0, 0.0), (0, 0.0)), threads=1):
        """threads = 1: the single-thread checker.  threads = 0 (all cores) or > 1: the OpenMP build, for timing only."""
        self.fp = fp
        self.L = L = lib() if threads == 1 else lib_omp()
        self.threads = 1 if threads == 1 else L.orc_set_threads(int(threads))
        self._keep = [np.ascontiguousarray(a) for a in (fp.q, fp.t, fp.cam, fp.Xw, fp.eP, fp.eL, fp.eDim, fp.meas, fp.omega)]
        q, t, cam, Xw, eP, eL, eDim, meas, omega = self._keep
        self.h = C.c_void_p(L.orc_create(fp.Pt, fp.Pf, fp.Lt, fp.Lf, _d(q), _d(t), _d(cam), _d(Xw), fp.E,
                                         eP.ctypes.data_as(_ip), eL.ctypes.data_as(_ip),
                                         eDim.ctypes.data_as(C.POINTER(C.c_uint8)), _d(meas), _d(omega)))
        for et, (kind, delta) in enumerate(robust):
            L.orc_set_robust_kernel(self.h, et, int(kind), float(delta))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def build_structure(self): self.L.orc_build_structure(self.h)
    def compute_errors(self): return self.L.orc_compute_errors(self.h)
    def build_system(self): self.L.orc_build_system(self.h)
    def max_diagonal(self): return self.L.orc_max_diagonal(self.h)
    def set_lambda(self, lam): self.L.orc_set_lambda(self.h, float(lam))
    def restore_diagonal(self): self.L.orc_restore_diagonal(self.h)
    def solve(self): return bool(self.L.orc_solve(self.h))
    def schur(self): self.L.orc_schur(self.h)
    def solve_reduced(self): return bool(self.L.orc_solve_reduced(self.h))
    def back_substitute(self): self.L.orc_back_substitute(self.h)

    def set_array(self, name, values):
        ids = dict(self.ARR, hsc=11)
        v = np.ascontiguousarray(values, dtype=np.float64)
        n = self.L.orc_set_array(self.h, ids[name], _d(v))
        assert n == v.size, (name, n, v.size)

    def hsc_values_raw(self):
        """Hsc values exactly as stored (col-major 6x6 blocks, flattened)."""
        nb = self.L.orc_hsc_nblocks(self.h)
        rp, ci = np.zeros(self.fp.Pf + 1, dtype=np.int32), np.zeros(nb, dtype=np.int32)
        v = np.zeros(nb * 36)
        self.L.orc_get_hsc(self.h, rp.ctypes.data_as(_ip), ci.ctypes.data_as(_ip), _d(v))
        return rp, ci, v
    def update(self): self.L.orc_update(self.h)
    def compute_scale(self, lam): return self.L.orc_compute_scale(self.h, float(lam))
    def push(self): self.L.orc_push(self.h)
    def pop(self): self.L.orc_pop(self.h)

    def optimize(self, niter):
        chi2, lam, trials = np.zeros(niter), np.zeros(niter), np.zeros(niter, dtype=np.int32)
        n = self.L.orc_optimize(self.h, niter, _d(chi2), _d(lam), trials.ctypes.data_as(_ip))
        return dict(chi2=chi2[:n], lambdas=lam[:n], trials=trials[:n])

    def chi_squares(self):
        out = np.zeros(self.fp.E)
        self.L.orc_chi_squares(self.h, _d(out))
        return out

    def state(self):
        q, t, X = np.zeros((self.fp.Pt, 4)), np.zeros((self.fp.Pt, 3)), np.zeros((self.fp.Lt, 3))
        self.L.orc_get_state(self.h, _d(q), _d(t), _d(X))
        return q, t, X

    def set_state(self, q, t, X):
        q, t, X = (np.ascontiguousarray(a, dtype=np.float64) for a in (q, t, X))
        self.L.orc_set_state(self.h, _d(q), _d(t), _d(X))

    def array(self, name):
        n = self.L.orc_get_array(self.h, self.ARR[name], None)
        out = np.zeros(n)
        self.L.orc_get_array(self.h, self.ARR[name], _d(out))
        return out

    def hsc(self):
        """Upper-triangular BSR of the reduced system: (rowptr, colind, values[nblk,6,6] row-major view)."""
        nb = self.L.orc_hsc_nblocks(self.h)
        rp, ci = np.zeros(self.fp.Pf + 1, dtype=np.int32), np.zeros(nb, dtype=np.int32)
        v = np.zeros(nb * 36)
        self.L.orc_get_hsc(self.h, rp.ctypes.data_as(_ip), ci.ctypes.data_as(_ip), _d(v))
        return rp, ci, v.reshape(nb, 6, 6).transpose(0, 2, 1).copy()   # col-major blocks -> [blk][row][col]
