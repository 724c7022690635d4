"""ctypes binding of the C ABI (include/cuba_hip.h -> csrc/libcuba_hip.so).

This is plumbing for tests and bench.py: every call goes straight through the C ABI to the HIP
kernels.  There is no CPU fallback -- if the shared library is missing or no GPU is visible the
constructor raises.  The method names mirror CudaBlockSolver's stage methods
(/root/reference/src/cuda_bundle_adjustment.cpp:115-562).
"""
from __future__ import annotations

import atexit
import ctypes as C
import os
import subprocess
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libcuba_hip.so")
LIB_PATH_F32 = os.path.join(CSRC, "libcuba_hip_f32.so")     # single-precision build (the reference's USE_FLOAT32 option)
HEADER = os.path.join(os.path.dirname(_HERE), "include", "cuba_hip.h")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)

ARRAY_IDS = dict(bp=0, bsc=1, xp=2, xl=3, lm_sys=4, hsc=5, state=6)
PROFILE_KEYS = (  # same strings as CudaBlockSolver::getTimeProfile (src/cuda_bundle_adjustment.cpp:545-562)
    "0: Initialize Optimizer", "1: Build Structure", "2: Compute Error", "3: Build System",
    "4: Schur Complement", "5: Symbolic Decomposition", "6: Numerical Decomposition", "7: Update Solution")


class CubaHipError(RuntimeError):
    pass


def build_library(force=False):
    """Compile csrc/*.hip for gfx950 (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))] + [HEADER]
    stale = not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale or not os.path.exists(LIB_PATH_F32):
        subprocess.check_call(["make", "-C", CSRC, "-s", "all"])
    return LIB_PATH


_libs = {}


def load_library(precision="f64"):
    path = {"f64": LIB_PATH, "f32": LIB_PATH_F32}[precision]
    path = os.environ.get("CUBA_HIP_LIB_" + precision.upper(), path)     # e.g. the stage-timestamp build (make libcuba_hip_trace.so)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise CubaHipError(f"{path} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
    if os.environ.get("CUBA_HIP_NO_TORCH") != "1":
        # PyTorch wheels bundle their own libamdhip64; whichever HIP runtime is loaded first serves the whole
        # process.  Loading torch's first keeps torch.cuda (streams, torch.distributed/RCCL) usable next to this
        # library; the C ABI itself does not depend on torch.
        try:
            import torch  # noqa: F401
        except Exception:      # pragma: no cover
            pass
    lib = C.CDLL(path)
    H = C.c_void_p
    sig = {
        "cuba_hip_create": [C.c_int, C.POINTER(H)],
        "cuba_hip_destroy": [H],
        "cuba_hip_set_stream": [H, C.c_void_p],
        "cuba_hip_set_option": [H, C.c_char_p, C.c_double],
        "cuba_hip_set_graph": [H, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_int, _ip, _ip, _u8p, _dp, _dp],
        "cuba_hip_set_graph_begin": [H, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_int, _ip, _ip, _u8p, _dp, _dp],
        "cuba_hip_set_graph_end": [H],
        "cuba_hip_set_graph_partition": [H, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_int, _ip, _ip, _u8p, _dp, _dp, C.c_int, C.c_int],
        "cuba_hip_set_robust_kernel": [H, C.c_int, C.c_int, C.c_double],
        "cuba_hip_build_structure": [H],
        "cuba_hip_compute_errors": [H, _dp],
        "cuba_hip_build_system": [H],
        "cuba_hip_max_diagonal": [H, _dp],
        "cuba_hip_set_lambda": [H, C.c_double],
        "cuba_hip_restore_diagonal": [H],
        "cuba_hip_schur": [H],
        "cuba_hip_schur_parts": [H, C.POINTER(C.c_int)],
        "cuba_hip_schur_part": [H, C.c_int, C.POINTER(C.c_size_t)],
        "cuba_hip_solve_reduced": [H, C.POINTER(C.c_int)],
        "cuba_hip_back_substitute": [H],
        "cuba_hip_solve": [H, C.POINTER(C.c_int)],
        "cuba_hip_update": [H],
        "cuba_hip_compute_scale": [H, C.c_double, _dp],
        "cuba_hip_push": [H],
        "cuba_hip_pop": [H],
        "cuba_hip_hint_unchanged": [H, C.c_int, C.c_int],
        "cuba_hip_chi_squares_begin": [H, _dp],
        "cuba_hip_chi_squares_end": [H],
        "cuba_hip_snapshot_state": [H],
        "cuba_hip_restore_state": [H],
        "cuba_hip_snapshot_state_slot": [H, C.c_int],
        "cuba_hip_restore_state_slot": [H, C.c_int],
        "cuba_hip_get_counter": [H, C.c_char_p, C.POINTER(C.c_int64)],
        "cuba_hip_optimize": [H, C.c_int, _dp, C.POINTER(C.c_int)],
        "cuba_hip_optimize_batch": [C.POINTER(H), C.c_int, C.c_int, _dp, C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "cuba_hip_get_solution": [H, _dp, _dp, _dp],
        "cuba_hip_set_solution": [H, _dp, _dp, _dp],
        "cuba_hip_chi_squares": [H, _dp],
        "cuba_hip_get_profile": [H, _dp],
        "cuba_hip_get_counters": [H, C.POINTER(C.c_int64)],
        "cuba_hip_get_hsc_structure": [H, _ip, _ip, C.POINTER(C.c_int)],
        "cuba_hip_get_pcg_history": [H, _ip, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)],
        "cuba_hip_get_array": [H, C.c_int, _dp, C.POINTER(C.c_size_t)],
        "cuba_hip_time_kernels": [H, C.c_int, _dp],
        "cuba_hip_set_partition": [H, C.c_int, C.c_int],
        "cuba_hip_assemble": [H],
        "cuba_hip_max_diagonal_parts": [H, _dp, _dp],
        "cuba_hip_compute_scale_parts": [H, C.c_double, _dp, _dp],
        "cuba_hip_device_pointer": [H, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)],
        "cuba_hip_reduction_buffer": [H, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)],
        "cuba_hip_get_stream": [H, C.POINTER(C.c_void_p)],
        "cuba_hip_begin_run": [H],
        "cuba_hip_get_sizes": [H, C.POINTER(C.c_int)],
        "cuba_hip_evaluate_device": [H, C.c_double, C.c_int, C.POINTER(C.c_void_p)],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.cuba_hip_host_alloc.argtypes = [C.c_size_t]
    lib.cuba_hip_host_alloc.restype = C.c_void_p
    lib.cuba_hip_host_free.argtypes = [C.c_void_p]
    lib.cuba_hip_host_free.restype = None
    lib.cuba_hip_debug_dense_inverse.argtypes = [C.c_int, C.c_int, _dp, _dp]
    lib.cuba_hip_debug_dense_inverse.restype = C.c_int
    lib.cuba_hip_debug_dense_solve.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, C.POINTER(C.c_int)]
    lib.cuba_hip_debug_dense_solve.restype = C.c_int
    lib.cuba_hip_debug_sparse_solve.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int32)]
    lib.cuba_hip_debug_sparse_solve.restype = C.c_int
    lib.cuba_hip_debug_sparse_plan.argtypes = [C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.c_int, C.POINTER(C.c_int32),
                                               C.c_size_t, C.POINTER(C.c_size_t)]
    lib.cuba_hip_debug_sparse_plan.restype = C.c_int
    lib.cuba_hip_last_error.argtypes = [H]
    lib.cuba_hip_last_error.restype = C.c_char_p
    lib.cuba_hip_version.restype = C.c_char_p
    lib.cuba_hip_scalar_size.restype = C.c_int
    _libs[path] = lib
    return lib


def _d(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def dense_inverse(A, precision="f64", device=0):
    """The library's blocked Gauss-Jordan inversion (coarse level of the preconditioner) applied to an SPD matrix: test hook."""
    A = np.asfortranarray(A, dtype=np.float64)
    out = np.zeros_like(A, order="F")
    rc = load_library(precision).cuba_hip_debug_dense_inverse(int(device), A.shape[0], _d(A), _d(out))
    if rc != 0:
        raise CubaHipError(f"cuba_hip_debug_dense_inverse failed with status {rc}")
    return np.array(out)


_live = weakref.WeakSet()


@atexit.register
def _close_all():
    # destroy device objects (streams, graphs, buffers) while the HIP runtime is still alive: handles that
    # survive until interpreter teardown would be destroyed after the runtime's own static destructors
    for s in list(_live):
        s.close()


def dense_solve(A, b, precision="f64", device=0, slack=-1, with_stats=False):
    """The library's exact reduced solve (sparse tile Cholesky on the matrix cores, csrc/ba_direct.hip) applied to a symmetric
    matrix whose order is a multiple of 6 (6 x 6 blocks that are identically zero stay out of the pattern): test hook.
    Returns (x, not_positive_definite[, {tile columns, tiles, levels, slack}])."""
    A = np.asfortranarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros_like(b)
    flag = C.c_int()
    stats = (C.c_int32 * 4)()
    rc = load_library(precision).cuba_hip_debug_sparse_solve(int(device), A.shape[0], _d(A), _d(b), _d(x), C.byref(flag), int(slack), stats)
    if rc != 0:
        raise CubaHipError(f"cuba_hip_debug_sparse_solve failed with status {rc}")
    if with_stats:
        return x, bool(flag.value), dict(zip(("tile_columns", "tiles", "levels", "slack"), (int(v) for v in stats)))
    return x, bool(flag.value)


SPARSE_PLAN_ARRAYS = ("header", "posOfSeg", "colPtr", "rowIdx", "gPtr", "gather", "lvlPtr", "lvlTiles", "lvlColPtr", "lvlCols", "blkTile")


def sparse_plan(row_ptr, col_ind, slack=-1, precision="f64"):
    """Symbolic phase of the exact reduced solve for an upper-triangular block pattern (host only, no device): dict of the plan's arrays
    (SparseCholPlan in csrc/ba_kernels.hpp)."""
    lib = load_library(precision)
    rp = np.ascontiguousarray(row_ptr, dtype=np.int32); ci = np.ascontiguousarray(col_ind, dtype=np.int32)
    ip = C.POINTER(C.c_int32)
    out = {}
    for which, name in enumerate(SPARSE_PLAN_ARRAYS):
        n = C.c_size_t()
        rc = lib.cuba_hip_debug_sparse_plan(len(rp) - 1, rp.ctypes.data_as(ip), ci.ctypes.data_as(ip), int(slack), which, None, 0, C.byref(n))
        if rc != 0:
            raise CubaHipError(f"cuba_hip_debug_sparse_plan failed with status {rc}")
        a = np.zeros(n.value, dtype=np.int32)
        rc = lib.cuba_hip_debug_sparse_plan(len(rp) - 1, rp.ctypes.data_as(ip), ci.ctypes.data_as(ip), int(slack), which, a.ctypes.data_as(ip), n.value, C.byref(n))
        if rc != 0:
            raise CubaHipError(f"cuba_hip_debug_sparse_plan failed with status {rc}")
        out[name] = a
    h = out.pop("header")
    out.update(T=int(h[0]), nTiles=int(h[1]), nLevels=int(h[2]), slack=int(h[3]), entries=int(h[4]), nblk=int(h[5]))
    return out


def optimize_batch(solvers, niter):
    """cuba_hip_optimize_batch: the LM runs of several HipSolver handles (one library, one device) in one launch chain.
    Returns ([chi2 per iteration of every handle], reduced solves that ran batched)."""
    n = len(solvers)
    lib = solvers[0].lib
    arr = (C.c_void_p * n)(*[s.h.value for s in solvers])
    chi2 = np.zeros((n, max(niter, 1)))
    done = (C.c_int * n)()
    batched = C.c_int()
    rc = lib.cuba_hip_optimize_batch(arr, n, int(niter), _d(chi2), done, C.byref(batched))
    if rc != 0:
        raise CubaHipError(f"cuba_hip_optimize_batch failed with status {rc}: {lib.cuba_hip_last_error(solvers[0].h).decode()}")
    return [chi2[i, :done[i]].copy() for i in range(n)], int(batched.value)


class HipSolver:
    """One bundle-adjustment problem on one GPU, driven through the C ABI."""

    def __init__(self, fp=None, robust=((0, 0.0), (0, 0.0)), device=0, stream=None, precision="f64", **options):
        self.lib = load_library(precision)
        self.scalar_size = self.lib.cuba_hip_scalar_size()
        self.h = C.c_void_p()
        rc = self.lib.cuba_hip_create(int(device), C.byref(self.h))
        if rc != 0:
            self.h = None
            raise CubaHipError(f"cuba_hip_create failed with status {rc} (4 = no HIP device visible)")
        if stream is not None:
            self._ck(self.lib.cuba_hip_set_stream(self.h, C.c_void_p(int(stream))))
        for k, v in options.items():
            self.set_option(k, v)
        self.fp = None
        _live.add(self)
        for et, (kind, delta) in enumerate(robust):
            self.set_robust_kernel(et, kind, delta)
        if fp is not None:
            self.set_graph(fp)

    def close(self):
        if getattr(self, "h", None):
            self.lib.cuba_hip_destroy(self.h)
            self.h = None

    __del__ = close

    def _ck(self, rc):
        if rc != 0:
            raise CubaHipError(f"status {rc}: {self.lib.cuba_hip_last_error(self.h).decode()}")

    def set_option(self, key, value):
        self._ck(self.lib.cuba_hip_set_option(self.h, key.encode(), float(value)))

    def set_robust_kernel(self, edge_type, kind, delta):
        self._ck(self.lib.cuba_hip_set_robust_kernel(self.h, int(edge_type), int(kind), float(delta)))

    def hint_unchanged(self, same_edges=True, same_values=False):
        """promise about the NEXT set_graph call only (cuba_hip_hint_unchanged)"""
        self._ck(self.lib.cuba_hip_hint_unchanged(self.h, int(bool(same_edges)), int(bool(same_values))))

    def set_graph(self, fp, two_step=False, landmark_range=None):
        """two_step: cuba_hip_set_graph_begin + cuba_hip_build_structure + cuba_hip_set_graph_end (what the C++ layer does: the
        measurements cross PCIe on a second stream while the structure analysis runs).
        landmark_range = (begin, end): cuba_hip_set_graph_partition -- the upload of one rank of a landmark partition, which sends the
        measurements and information of its own landmarks' edges only."""
        self.fp = fp
        q, t, cam, Xw = (np.ascontiguousarray(a, dtype=np.float64) for a in (fp.q, fp.t, fp.cam, fp.Xw))
        eP = np.ascontiguousarray(fp.eP, dtype=np.int32)
        eL = np.ascontiguousarray(fp.eL, dtype=np.int32)
        eD = np.ascontiguousarray(fp.eDim, dtype=np.uint8)
        meas = np.ascontiguousarray(fp.meas, dtype=np.float64)
        om = np.ascontiguousarray(fp.omega, dtype=np.float64)
        if landmark_range is not None:
            self._ck(self.lib.cuba_hip_set_graph_partition(self.h, fp.Pt, fp.Pf, fp.Lt, fp.Lf, _d(q), _d(t), _d(cam), _d(Xw), len(eP),
                     eP.ctypes.data_as(_ip), eL.ctypes.data_as(_ip), eD.ctypes.data_as(_u8p), _d(meas), _d(om),
                     int(landmark_range[0]), int(landmark_range[1])))
            return
        fn = self.lib.cuba_hip_set_graph_begin if two_step else self.lib.cuba_hip_set_graph
        self._ck(fn(self.h, fp.Pt, fp.Pf, fp.Lt, fp.Lf, _d(q), _d(t), _d(cam), _d(Xw), len(eP),
                    eP.ctypes.data_as(_ip), eL.ctypes.data_as(_ip), eD.ctypes.data_as(_u8p), _d(meas), _d(om)))
        if two_step == "begin_only":                                 # (test hook: the arrays are kept alive by the caller of this method)
            self._pending_upload = (meas, om)
        elif two_step:
            self.build_structure()
            self._ck(self.lib.cuba_hip_set_graph_end(self.h))        # (meas / om stay referenced until here)

    # ---- stages --------------------------------------------------------------------------------
    def build_structure(self): self._ck(self.lib.cuba_hip_build_structure(self.h))

    def compute_errors(self):
        v = C.c_double()
        self._ck(self.lib.cuba_hip_compute_errors(self.h, C.byref(v)))
        return v.value

    def build_system(self): self._ck(self.lib.cuba_hip_build_system(self.h))

    def max_diagonal(self):
        v = C.c_double()
        self._ck(self.lib.cuba_hip_max_diagonal(self.h, C.byref(v)))
        return v.value

    def set_lambda(self, lam): self._ck(self.lib.cuba_hip_set_lambda(self.h, float(lam)))
    def restore_diagonal(self): self._ck(self.lib.cuba_hip_restore_diagonal(self.h))
    def schur(self): self._ck(self.lib.cuba_hip_schur(self.h))

    def schur_parts(self):
        n = C.c_int()
        self._ck(self.lib.cuba_hip_schur_parts(self.h, C.byref(n)))
        return n.value

    def schur_part(self, part):
        """Part `part` of schur(); returns ((offset, count), (offset, count)): the ranges of the reduction buffer it completes."""
        r = (C.c_size_t * 4)()
        self._ck(self.lib.cuba_hip_schur_part(self.h, int(part), r))
        return (int(r[0]), int(r[1])), (int(r[2]), int(r[3]))

    def solve_reduced(self):
        ok = C.c_int()
        self._ck(self.lib.cuba_hip_solve_reduced(self.h, C.byref(ok)))
        return bool(ok.value)

    def back_substitute(self): self._ck(self.lib.cuba_hip_back_substitute(self.h))

    def solve(self):
        ok = C.c_int()
        self._ck(self.lib.cuba_hip_solve(self.h, C.byref(ok)))
        return bool(ok.value)

    def update(self): self._ck(self.lib.cuba_hip_update(self.h))

    def compute_scale(self, lam):
        v = C.c_double()
        self._ck(self.lib.cuba_hip_compute_scale(self.h, float(lam), C.byref(v)))
        return v.value

    def push(self): self._ck(self.lib.cuba_hip_push(self.h))
    def pop(self): self._ck(self.lib.cuba_hip_pop(self.h))
    def snapshot_state(self, slot=0): self._ck(self.lib.cuba_hip_snapshot_state_slot(self.h, int(slot)))
    def restore_state(self, slot=0): self._ck(self.lib.cuba_hip_restore_state_slot(self.h, int(slot)))

    def optimize(self, niter):
        chi2 = np.zeros(max(niter, 1))
        n = C.c_int()
        self._ck(self.lib.cuba_hip_optimize(self.h, int(niter), _d(chi2), C.byref(n)))
        return dict(chi2=chi2[:n.value])

    # ---- results -------------------------------------------------------------------------------
    def state(self):
        q, t, X = np.zeros((self.fp.Pt, 4)), np.zeros((self.fp.Pt, 3)), np.zeros((self.fp.Lt, 3))
        self._ck(self.lib.cuba_hip_get_solution(self.h, _d(q), _d(t), _d(X)))
        return q, t, X

    def set_state(self, q, t, X):
        q, t, X = (np.ascontiguousarray(a, dtype=np.float64) for a in (q, t, X))
        self._ck(self.lib.cuba_hip_set_solution(self.h, _d(q), _d(t), _d(X)))

    def chi_squares(self):
        out = np.zeros(self.fp.E)
        self._ck(self.lib.cuba_hip_chi_squares(self.h, _d(out)))
        return out

    def chi_squares_two_step(self):
        """cuba_hip_chi_squares_begin / _end (the C++ layer does its write-back between the two)"""
        out = np.zeros(self.fp.E)
        self._ck(self.lib.cuba_hip_chi_squares_begin(self.h, _d(out)))
        self._ck(self.lib.cuba_hip_chi_squares_end(self.h))
        return out

    def profile(self):
        out = np.zeros(8)
        self._ck(self.lib.cuba_hip_get_profile(self.h, _d(out)))
        return dict(zip(PROFILE_KEYS, out.tolist()))

    def counters(self):
        c = (C.c_int64 * 8)()
        self._ck(self.lib.cuba_hip_get_counters(self.h, c))
        return dict(pcg_iterations=int(c[0]), lm_trials=int(c[1]), hsc_blocks=int(c[2]), schur_products=int(c[3]),
                    coarse_refreshes=int(c[4]), pcg_host_looks=int(c[5]), pcg_iterations_enqueued=int(c[6]), coarse_dim=int(c[7]))

    def counter(self, name):
        """one counter by name (cuba_hip_get_counter), e.g. "coarse_inline_inversions", "pcg_graph_instantiations"."""
        v = C.c_int64()
        self._ck(self.lib.cuba_hip_get_counter(self.h, name.encode(), C.byref(v)))
        return int(v.value)

    def pcg_history(self):
        """(iterations per reduced solve since set_graph [negative = stopped at max_iter], number of unconverged solves)."""
        n, bad = C.c_int(), C.c_int64()
        self._ck(self.lib.cuba_hip_get_pcg_history(self.h, None, 0, C.byref(n), C.byref(bad)))
        it = np.zeros(max(n.value, 1), dtype=np.int32)
        self._ck(self.lib.cuba_hip_get_pcg_history(self.h, it.ctypes.data_as(_ip), n.value, C.byref(n), C.byref(bad)))
        return it[:n.value], int(bad.value)

    def array(self, name):
        n = C.c_size_t()
        self._ck(self.lib.cuba_hip_get_array(self.h, ARRAY_IDS[name], None, C.byref(n)))
        out = np.zeros(n.value)
        if n.value:
            self._ck(self.lib.cuba_hip_get_array(self.h, ARRAY_IDS[name], _d(out), C.byref(n)))
        return out

    def hsc_structure(self):
        nb = C.c_int()
        self._ck(self.lib.cuba_hip_get_hsc_structure(self.h, None, None, C.byref(nb)))
        rp, ci = np.zeros(self.fp.Pf + 1, dtype=np.int32), np.zeros(nb.value, dtype=np.int32)
        self._ck(self.lib.cuba_hip_get_hsc_structure(self.h, rp.ctypes.data_as(_ip), ci.ctypes.data_as(_ip), C.byref(nb)))
        return rp, ci

    def hsc(self):
        """(rowptr, colind, values[nblk,6,6] indexed [blk][row][col]) of the upper-triangular BSR."""
        rp, ci = self.hsc_structure()
        v = self.array("hsc").reshape(len(ci), 6, 6).transpose(0, 2, 1).copy()
        return rp, ci, v

    def time_kernels(self, reps=20):
        out = np.zeros(7)
        self._ck(self.lib.cuba_hip_time_kernels(self.h, int(reps), _d(out)))
        return dict(zip(("residual_chi2", "linearize_schur", "pcg_spmv", "pcg_update", "back_substitute", "pcg_precond",
                         "coarse_setup"), out.tolist()))

    # ---- landmark-partitioned multi-GPU hooks ---------------------------------------------------
    def set_partition(self, lm_begin, lm_end):
        self._ck(self.lib.cuba_hip_set_partition(self.h, int(lm_begin), int(lm_end)))

    def assemble(self): self._ck(self.lib.cuba_hip_assemble(self.h))

    def max_diagonal_parts(self):
        a, b = C.c_double(), C.c_double()
        self._ck(self.lib.cuba_hip_max_diagonal_parts(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def compute_scale_parts(self, lam):
        a, b = C.c_double(), C.c_double()
        self._ck(self.lib.cuba_hip_compute_scale_parts(self.h, float(lam), C.byref(a), C.byref(b)))
        return a.value, b.value

    def device_pointer(self, name):
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(self.lib.cuba_hip_device_pointer(self.h, ARRAY_IDS[name], C.byref(p), C.byref(n)))
        return p.value, n.value

    def reduction_buffer(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(self.lib.cuba_hip_reduction_buffer(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value
