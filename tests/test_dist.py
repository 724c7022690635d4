"""Landmark-partitioned multi-process LM (cuda-bundle-adjustment_amd/dist.py).
CPU: world_size-2 gloo run of the driver + communicator on an oracle-backed stand-in backend.
GPU: the HIP partition path (cuba_hip_set_partition & friends) with two emulated ranks on one device."""
import json
import socket
import threading

import numpy as np
import pytest

from conftest import RK_HUBER
from cuba_amd.dist import ThreadComm, landmark_ranges, partitioned_optimize
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_ba

GRAPH = dict(P=40, L=600, E=2400, seed=1)


def test_landmark_ranges_cover_and_balance():
    fp = flatten(synth_ba(**GRAPH))
    for world in (1, 2, 3, 8):
        r = landmark_ranges(fp.eL, fp.Lt, world)
        assert r[0][0] == 0 and r[-1][1] == fp.Lt and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        cnt = [int(((fp.eL >= lo) & (fp.eL < hi)).sum()) for lo, hi in r]
        assert sum(cnt) == fp.E and max(cnt) - min(cnt) <= 64


def test_partitioned_lm_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    from dist_helpers import gloo_rank_main
    from oracle.oracle import OracleSolver
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    out = str(tmp_path / "rank0.json")
    mp.spawn(gloo_rank_main, args=(2, port, out, GRAPH, RK_HUBER, 6), nprocs=2, join=True)
    got = json.load(open(out))
    fp = flatten(synth_ba(**GRAPH))
    ref = OracleSolver(fp, RK_HUBER); r = ref.optimize(6)
    assert len(got["chi2"]) == len(r["chi2"])
    assert np.allclose(got["chi2"], r["chi2"], rtol=1e-9)
    q, t, X = ref.state()
    assert np.allclose(got["X"], X, atol=1e-8) and np.allclose(got["t"], t, atol=1e-8)


def test_thread_comm_collectives():
    comms = ThreadComm.create(3)
    res = [None] * 3

    def work(c):
        a = np.full(4, float(c.rank + 1)); c.allreduce_sum_(a)
        b = np.full(2, float(c.rank)); c.bcast_(b, 1)
        res[c.rank] = (a.copy(), b.copy(), c.sum(c.rank), c.max(c.rank), c.min(c.rank))
    th = [threading.Thread(target=work, args=(c,)) for c in comms]
    [t.start() for t in th]; [t.join() for t in th]
    for a, b, s, mx, mn in res:
        assert np.all(a == 6) and np.all(b == 1) and (s, mx, mn) == (3.0, 2.0, 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_hip_path_emulated_ranks(world):
    """Two/three solver handles on one GPU act as ranks (ThreadComm): same chi2 trajectory and estimates as
    the single-handle solve."""
    from cuba_amd.capi import HipSolver
    from cuba_amd.dist import HipPartitionBackend
    fp = flatten(synth_ba(120, 6000, 24000, seed=9))
    single = HipSolver(fp, RK_HUBER)
    want = single.optimize(8)["chi2"]
    q1, t1, X1 = single.state()
    comms = ThreadComm.create(world)
    out = [None] * world
    err = []

    def work(c):
        try:
            be = HipPartitionBackend(HipSolver(fp, RK_HUBER), fp, c.rank, world)
            chi2 = partitioned_optimize(be, c, 8)
            out[c.rank] = (chi2, be.gather_solution(c))
        except Exception as e:   # pragma: no cover
            err.append(e)
            c.s.barrier.abort()
    th = [threading.Thread(target=work, args=(c,)) for c in comms]
    [t.start() for t in th]; [t.join() for t in th]
    assert not err, err
    for chi2, (q, t, X) in out:
        assert len(chi2) == len(want) and np.all(np.abs(chi2 - want) <= 1e-8 * want)
        assert np.abs(X - X1).max() < 1e-6 and np.abs(t - t1).max() < 1e-6 and np.abs(q - q1).max() < 1e-8
