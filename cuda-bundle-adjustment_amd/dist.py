"""Landmark-partitioned multi-GPU Levenberg-Marquardt (BASELINE.json config 5).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" on CPU for tests).
Every rank holds the whole graph but evaluates only its contiguous landmark range
(`cuba_hip_set_partition`): poses, lambda and the reduced solve are replicated, and per LM trial there is
exactly ONE large exchange -- the sum of the reduction buffer [Hsc | bsc | bp] -- plus a few scalars
(chi2, landmark part of the gain-ratio denominator, the solver's ok flag) and a broadcast of the 6*Pf pose
increments so that replicas cannot drift apart by rounding.  The control flow is that of
CudaBundleAdjustmentImpl::optimize (/root/reference/src/cuda_bundle_adjustment.cpp:793-857); the reference
itself is single-GPU and has no counterpart of this file.

The driver is written against a small `backend` interface so that the same code runs on the HIP path
(`HipPartitionBackend`) and -- in the CPU-only gloo tests -- on a stand-in backend.
"""
from __future__ import annotations

import threading

import numpy as np


# ---------------------------------------------------------------------------------------------------
# landmark ranges
# ---------------------------------------------------------------------------------------------------
def landmark_ranges(edge_landmark, Lt, world):
    """Contiguous landmark ranges [begin, end) per rank, balanced by edge count."""
    cnt = np.bincount(np.asarray(edge_landmark), minlength=Lt).astype(np.int64)
    cum = np.concatenate([[0], np.cumsum(cnt)])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(cum, total * r / world, side="left")))
    cuts.append(Lt)
    cuts = np.maximum.accumulate(np.minimum(cuts, Lt))
    return [(int(cuts[r]), int(cuts[r + 1])) for r in range(world)]


# ---------------------------------------------------------------------------------------------------
# communicators
# ---------------------------------------------------------------------------------------------------
class TorchComm:
    """torch.distributed process group (nccl on GPUs, gloo on CPU). Accepts torch tensors or numpy arrays."""

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")

    def _stage(self, a):
        """tensor the backend can work on (+ whether the result has to be copied back into `a`)"""
        if isinstance(a, np.ndarray):
            t = self.torch.from_numpy(a)
            return (t, False) if self.device == "cpu" else (t.to(self.device), True)
        if self.device == "cpu" and a.is_cuda:          # gloo dry runs with device buffers: stage through the host
            return a.cpu(), True
        return a, False

    def _unstage(self, a, t):
        if isinstance(a, np.ndarray):
            a[...] = t.cpu().numpy()
        else:
            a.copy_(t)

    def allreduce_sum_(self, a):
        t, back = self._stage(a)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        if back:
            self._unstage(a, t)
        return a

    def allreduce_max_(self, a):
        t, back = self._stage(a)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        if back:
            self._unstage(a, t)
        return a

    def bcast_(self, a, root=0):
        t, back = self._stage(a)
        self.dist.broadcast(t, src=root, group=self.group)
        if back:
            self._unstage(a, t)
        return a

    def _scalar(self, v, op):
        t = self.torch.tensor([float(v)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=op, group=self.group)
        return float(t.item())

    def sum(self, v): return self._scalar(v, self.dist.ReduceOp.SUM)
    def max(self, v): return self._scalar(v, self.dist.ReduceOp.MAX)
    def min(self, v): return self._scalar(v, self.dist.ReduceOp.MIN)


class ThreadComm:
    """In-process emulation of `world` ranks as threads (one GPU, several solver handles): lets the GPU box
    exercise the partitioned HIP path without a multi-GPU node.  Works on numpy arrays and torch tensors."""

    class _Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, shared, rank):
        self.s, self.rank, self.world = shared, rank, shared.world

    @staticmethod
    def create(world):
        sh = ThreadComm._Shared(world)
        return [ThreadComm(sh, r) for r in range(world)]

    def _gather(self, v):
        self.s.slots[self.rank] = v
        self.s.barrier.wait()
        vals = list(self.s.slots)
        self.s.barrier.wait()
        return vals

    def allreduce_sum_(self, a):
        self.s.slots[self.rank] = a
        self.s.barrier.wait()
        if self.rank == 0:
            for other in self.s.slots[1:]:
                self.s.slots[0] += other
        self.s.barrier.wait()
        if self.rank != 0:
            src = self.s.slots[0]
            if isinstance(a, np.ndarray):
                a[...] = src
            else:
                a.copy_(src)
        self.s.barrier.wait()
        return a

    def bcast_(self, a, root=0):
        self.s.slots[self.rank] = a
        self.s.barrier.wait()
        if self.rank != root:
            src = self.s.slots[root]
            if isinstance(a, np.ndarray):
                a[...] = src
            else:
                a.copy_(src)
        self.s.barrier.wait()
        return a

    def allreduce_max_(self, a):
        self.s.slots[self.rank] = a
        self.s.barrier.wait()
        if self.rank == 0:
            import torch
            for other in self.s.slots[1:]:
                if isinstance(a, np.ndarray):
                    np.maximum(self.s.slots[0], other, out=self.s.slots[0])
                else:
                    torch.maximum(self.s.slots[0], other, out=self.s.slots[0])
        self.s.barrier.wait()
        if self.rank != 0:
            src = self.s.slots[0]
            if isinstance(a, np.ndarray):
                a[...] = src
            else:
                a.copy_(src)
        self.s.barrier.wait()
        return a

    def sum(self, v): return float(sum(self._gather(float(v))))
    def max(self, v): return float(max(self._gather(float(v))))
    def min(self, v): return float(min(self._gather(float(v))))


# ---------------------------------------------------------------------------------------------------
# HIP backend: one solver handle restricted to a landmark range
# ---------------------------------------------------------------------------------------------------
class _DeviceArray:
    def __init__(self, ptr, n, itemsize=8):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f%d" % itemsize, "data": (ptr, False), "version": 2}


class HipPartitionBackend:
    def __init__(self, solver, fp, rank, world):
        import torch
        self.h, self.fp = solver, fp
        self.range = landmark_ranges(fp.eL, fp.Lt, world)[rank]
        solver.set_partition(*self.range)
        solver.build_structure()
        ptr, n = solver.reduction_buffer()
        self.red = torch.as_tensor(_DeviceArray(ptr, n, solver.scalar_size), device="cuda")
        ptr, n = solver.device_pointer("xp")
        self.xp = torch.as_tensor(_DeviceArray(ptr, n, solver.scalar_size), device="cuda") if n else None
        self._torch = torch

    def _sync(self):
        self._torch.cuda.synchronize()

    def compute_errors(self): return self.h.compute_errors()
    def assemble(self): self.h.assemble()
    def max_diagonal_parts(self): return self.h.max_diagonal_parts()
    def set_lambda(self, lam): self.h.set_lambda(lam)
    def schur(self): self.h.schur()
    def solve_reduced(self): return self.h.solve_reduced()
    def back_substitute(self): self.h.back_substitute()
    def update(self): self.h.update()
    def compute_scale_parts(self, lam): return self.h.compute_scale_parts(lam)
    def push(self): self.h.push()
    def pop(self): self.h.pop()

    def allreduce_system(self, comm):
        self._sync()                       # kernels of this handle's stream are done before the collective reads
        comm.allreduce_sum_(self.red)
        self._sync()

    def bcast_increments(self, comm):
        if self.xp is not None:
            self._sync()
            comm.bcast_(self.xp, 0)
            self._sync()

    def gather_solution(self, comm):
        """Full (q, t, Xw) on every rank: each rank contributes the landmarks it owns."""
        q, t, X = self.h.state()
        lo, hi = self.range
        mask = np.zeros_like(X)
        mask[lo:hi] = X[lo:hi]
        comm.allreduce_sum_(mask)
        return q, t, mask


# ---------------------------------------------------------------------------------------------------
# the driver
# ---------------------------------------------------------------------------------------------------
def partitioned_optimize(backend, comm, niterations, maxq=10, tau=1e-5):
    """Levenberg-Marquardt over a landmark-partitioned graph. Returns per-iteration chi2 (identical on all ranks)."""
    nu, lam = 2.0, 0.0
    stats = []
    for it in range(niterations):
        F = comm.sum(backend.compute_errors())
        if it == 0:
            backend.assemble()
            backend.allreduce_system(comm)
            pose_part, lm_part = backend.max_diagonal_parts()
            lam = tau * max(pose_part, comm.max(lm_part))
        q, rho = 0, -1.0
        while q < maxq and rho < 0:
            backend.push()
            backend.set_lambda(lam)
            backend.schur()
            backend.allreduce_system(comm)                       # the one large exchange of the trial
            ok = comm.min(1.0 if backend.solve_reduced() else 0.0) > 0.5
            if ok:
                backend.bcast_increments(comm)
                backend.back_substitute()
                backend.update()
            Fhat = comm.sum(backend.compute_errors())
            scale = 1e-3
            if ok:
                sp, sl = backend.compute_scale_parts(lam)
                scale += sp + comm.sum(sl)
            rho = (F - Fhat) / scale if ok else -1.0
            if rho > 0:
                lam *= max(1.0 / 3, min(1 - (2 * rho - 1) ** 3, 2.0 / 3))
                nu = 2.0
                F = Fhat
                break
            lam *= nu
            nu *= 2
            backend.pop()
            q += 1
        stats.append(F)
        if q == maxq or rho <= 0 or not np.isfinite(lam):
            break
    return np.array(stats)


# ---------------------------------------------------------------------------------------------------
# native driver (libcuba_hip_dist.so, include/cuba_hip_dist.h): the same loop in C++, RCCL collectives enqueued on the
# solver's stream.  This class is ctypes plumbing only.
# ---------------------------------------------------------------------------------------------------
_dist_libs = {}


def load_dist_library(precision="f64"):
    import ctypes as C
    import os
    from . import capi
    capi.load_library(precision)                     # the solver library (and torch's HIP runtime) first
    path = os.path.join(capi.CSRC, "libcuba_hip_dist.so" if precision == "f64" else "libcuba_hip_dist_f32.so")
    if path in _dist_libs:
        return _dist_libs[path]
    if not os.path.exists(path):
        raise capi.CubaHipError(f"{path} is missing: run __graft_entry__.build()")
    lib = C.CDLL(path)
    H, D = C.c_void_p, C.c_void_p
    lib.cuba_hip_dist_unique_id.argtypes = [C.c_void_p]
    lib.cuba_hip_dist_create_rccl.argtypes = [H, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(D)]
    lib.cuba_hip_dist_attach_rccl.argtypes = [H, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(D)]
    lib.cuba_hip_dist_create_custom.argtypes = [H, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(D)]
    lib.cuba_hip_dist_optimize.argtypes = [D, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.cuba_hip_dist_complete_solution.argtypes = [D]
    lib.cuba_hip_dist_get_counters.argtypes = [D, C.POINTER(C.c_longlong)]
    lib.cuba_hip_dist_reduction_parts.argtypes = [D, C.POINTER(C.c_int), C.POINTER(C.c_longlong)]
    lib.cuba_hip_dist_destroy.argtypes = [D]
    lib.cuba_hip_dist_last_error.argtypes = [D]
    lib.cuba_hip_dist_last_error.restype = C.c_char_p
    _dist_libs[path] = lib
    return lib


def rccl_unique_id(precision="f64"):
    """128-byte RCCL unique id (rank 0 makes it, every rank passes it to NativeDist)."""
    import ctypes as C
    buf = C.create_string_buffer(128)
    rc = load_dist_library(precision).cuba_hip_dist_unique_id(buf)
    if rc != 0:
        raise RuntimeError(f"cuba_hip_dist_unique_id failed with status {rc}")
    return bytes(buf.raw)


class NativeDist:
    """Landmark-partitioned LM through the native driver.  Exactly one of `unique_id` (a new RCCL communicator) or `comm`
    (an object with allreduce_sum_ / allreduce_max_ on torch device tensors: ThreadComm for ranks-as-threads on one GPU,
    TorchComm for a torch.distributed group) selects the collectives."""

    def __init__(self, solver, fp, rank, world, unique_id=None, comm=None, precision="f64"):
        import ctypes as C
        self.lib = load_dist_library(precision)
        self.solver, self.rank, self.world = solver, rank, world
        self.range = landmark_ranges(fp.eL, fp.Lt, world)[rank]
        self.h = C.c_void_p()
        self._keep = None
        if (unique_id is None) == (comm is None):
            raise ValueError("pass either unique_id or comm")
        if unique_id is not None:
            buf = C.create_string_buffer(bytes(unique_id), 128)
            rc = self.lib.cuba_hip_dist_create_rccl(solver.h, buf, rank, world, self.range[0], self.range[1], C.byref(self.h))
        else:
            import torch
            FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)

            def make(op):
                def fn(ctx, buf, count, scalar_size, stream):
                    try:
                        torch.cuda.synchronize()                      # everything the solver enqueued so far
                        t = torch.as_tensor(_DeviceArray(buf, count, scalar_size), device="cuda")
                        getattr(comm, op)(t)
                        torch.cuda.synchronize()
                        return 0
                    except Exception:                                 # pragma: no cover
                        import traceback
                        traceback.print_exc()
                        return 1
                return FN(fn)

            class Ops(C.Structure):
                _fields_ = [("ctx", C.c_void_p), ("allreduce_sum", FN), ("allreduce_max", FN)]
            cb_sum, cb_max = make("allreduce_sum_"), make("allreduce_max_")
            ops = Ops(None, cb_sum, cb_max)
            self._keep = (cb_sum, cb_max, ops)                       # the C side calls these for the driver's lifetime
            rc = self.lib.cuba_hip_dist_create_custom(solver.h, C.byref(ops), rank, world, self.range[0], self.range[1], C.byref(self.h))
        if rc != 0:
            self.h = None
            raise RuntimeError(f"native multi-GPU driver: create failed with status {rc} ({solver.lib.cuba_hip_last_error(solver.h).decode()})")

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError(f"native multi-GPU driver: status {rc}: {self.lib.cuba_hip_dist_last_error(self.h).decode()}")

    def optimize(self, niter):
        import ctypes as C
        chi2 = np.zeros(max(niter, 1))
        n = C.c_int()
        self._ck(self.lib.cuba_hip_dist_optimize(self.h, int(niter), chi2.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n)))
        return chi2[:n.value]

    def complete_solution(self):
        """Afterwards solver.state() is the full solution on every rank."""
        self._ck(self.lib.cuba_hip_dist_complete_solution(self.h))
        return self.solver.state()

    def counters(self):
        import ctypes as C
        c = (C.c_longlong * 4)()
        self._ck(self.lib.cuba_hip_dist_get_counters(self.h, c))
        return dict(large_allreduces=int(c[0]), small_allreduces=int(c[1]), large_elements=int(c[2]), lm_trials=int(c[3]))

    def reduction_parts(self):
        """(parts the per-trial sum is issued in, all-reduces issued under a later part of the Schur pass so far)"""
        import ctypes as C
        n, k = C.c_int(), C.c_longlong()
        self._ck(self.lib.cuba_hip_dist_reduction_parts(self.h, C.byref(n), C.byref(k)))
        return n.value, int(k.value)

    def close(self):
        if getattr(self, "h", None):
            self.lib.cuba_hip_dist_destroy(self.h)
            self.h = None

    __del__ = close
