"""Test-only pieces for the landmark-partitioned driver: an oracle-backed stand-in backend (CPU) and the
entry point each gloo rank runs.  The product path (dist.HipPartitionBackend) never imports this."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class OraclePartitionBackend:
    """Full-graph oracle whose edges outside the rank's landmark range carry zero information: they then
    contribute nothing to chi2, H or b, while the Hsc pattern stays global -- the same contract as
    cuba_hip_set_partition."""

    def __init__(self, fp, robust, rank, world):
        import copy
        from cuba_amd.dist import landmark_ranges
        from oracle.oracle import OracleSolver
        self.range = landmark_ranges(fp.eL, fp.Lt, world)[rank]
        lo, hi = self.range
        local = copy.copy(fp)
        local.omega = np.where((fp.eL >= lo) & (fp.eL < hi), fp.omega, 0.0)
        self.fp, self.rank = fp, rank
        self.o = OracleSolver(local, robust)
        self.o.build_structure()
        self.lam = 0.0
        self.stage = None

    def compute_errors(self): return self.o.compute_errors()

    def assemble(self):
        self.o.build_system()
        self.stage = "assemble"

    def allreduce_system(self, comm):
        names = ("Hpp", "bp") if self.stage == "assemble" else ("bsc", "bp")
        for n in names:
            a = self.o.array(n); comm.allreduce_sum_(a); self.o.set_array(n, a)
        if self.stage == "schur":
            _, _, v = self.o.hsc_values_raw(); comm.allreduce_sum_(v); self.o.set_array("hsc", v)

    def max_diagonal_parts(self):
        Pf, Lf = self.fp.Pf, self.fp.Lf
        hpp = self.o.array("Hpp").reshape(Pf, 6, 6); hll = self.o.array("Hll").reshape(Lf, 3, 3)
        dp = max(0.0, max((np.diag(b).max() for b in hpp), default=0.0))
        dl = max(0.0, max((np.diag(b).max() for b in hll), default=0.0))
        return dp, dl

    def set_lambda(self, lam): self.lam = lam

    def schur(self):
        self.o.build_system()
        self.o.set_lambda(self.lam)
        if self.rank != 0:                      # lambda must enter the summed Hpp exactly once
            hpp = self.o.array("Hpp").reshape(self.fp.Pf, 6, 6)
            for k in range(6):
                hpp[:, k, k] -= self.lam
            self.o.set_array("Hpp", hpp)
        self.o.schur()
        self.stage = "schur"

    def solve_reduced(self): return self.o.solve_reduced()

    def bcast_increments(self, comm):
        a = self.o.array("xp"); comm.bcast_(a, 0); self.o.set_array("xp", a)

    def back_substitute(self): self.o.back_substitute()
    def update(self): self.o.update()

    def compute_scale_parts(self, lam):
        xp, bp, xl, bl = (self.o.array(n) for n in ("xp", "bp", "xl", "bl"))
        lo, hi = self.range
        m = np.zeros(self.fp.Lf * 3, bool); m[3 * min(lo, self.fp.Lf):3 * min(hi, self.fp.Lf)] = True
        return float(xp @ (lam * xp + bp)), float(xl[m] @ (lam * xl[m] + bl[m]))

    def push(self): self.o.push()

    def pop(self):
        self.o.pop()
        self.o.compute_errors()                 # refresh the stored residuals at the restored estimate

    def gather_solution(self, comm):
        q, t, X = self.o.state()
        lo, hi = self.range
        mask = np.zeros_like(X); mask[lo:hi] = X[lo:hi]
        comm.allreduce_sum_(mask)
        return q, t, mask


def gloo_rank_main(rank, world, port, out_path, graph_args, robust, iters):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cuba_amd.dist import TorchComm, partitioned_optimize
    from cuba_amd.graph import flatten
    from cuba_amd.synth import synth_ba
    fp = flatten(synth_ba(**graph_args))
    be = OraclePartitionBackend(fp, robust, rank, world)
    comm = TorchComm()
    chi2 = partitioned_optimize(be, comm, iters)
    q, t, X = be.gather_solution(comm)
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump({"chi2": chi2.tolist(), "q": q.tolist(), "t": t.tolist(), "X": X.tolist()}, f)
    dist.barrier()
    dist.destroy_process_group()


def rccl_rank_main(rank, world, port, out_dir, graph_args, robust, iters, precision="f64"):
    """One rank of a REAL multi-rank RCCL run of the native driver (tests/test_dist.py, needs `world` GPUs): device = rank, the
    128-byte RCCL unique id travels through a gloo group (CPU), the communicator itself is the library's own
    (cuba_hip_dist_create_rccl -> ncclCommInitRank) and every collective runs in-stream on the solver's stream."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cuba_amd.capi import HipSolver
    from cuba_amd.dist import NativeDist, rccl_unique_id
    from cuba_amd.graph import flatten
    from cuba_amd.synth import synth_ba
    fp = flatten(synth_ba(**graph_args))
    ids = [rccl_unique_id(precision) if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    h = HipSolver(fp, robust, device=rank, precision=precision)
    d = NativeDist(h, fp, rank, world, unique_id=ids[0], precision=precision)
    chi2 = d.optimize(iters)
    q, t, X = d.complete_solution()
    c = d.counters()
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({"chi2": chi2.tolist(), "q": q.tolist(), "t": t.tolist(), "X": X.tolist(), "counters": {k: int(v) for k, v in c.items()}}, f)
    d.close(); h.close()
    dist.barrier()
    dist.destroy_process_group()
