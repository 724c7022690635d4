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


def test_native_driver_library_exports_every_declared_symbol():
    """CPU: libcuba_hip_dist.so loads and exports what include/cuba_hip_dist.h declares (no compute without a GPU)."""
    import os
    import re
    from conftest import ROOT
    from cuba_amd.dist import load_dist_library
    lib = load_dist_library()
    header = open(os.path.join(ROOT, "include", "cuba_hip_dist.h")).read()
    names = set(re.findall(r"\b(cuba_hip_dist_\w+)\s*\(", header))
    assert len(names) >= 9
    for n in names:
        assert hasattr(lib, n), n


def _values_only_inside(fp, rng):
    """A copy of the graph whose measurements / information are NaN for every edge outside the landmark range: what a rank that holds
    only its own observations could pass to cuba_hip_set_graph_partition."""
    import copy
    lo, hi = rng
    own = (fp.eL >= lo) & (fp.eL < hi)
    g = copy.copy(fp)
    g.meas = np.where(own[:, None], fp.meas.reshape(len(own), -1), np.nan).reshape(fp.meas.shape)
    g.omega = np.where(own, fp.omega, np.nan)
    return g


def _run_native_ranks(fp, world, iters, make_comm_args, precision="f64", parts_out=None, ranged=False, **options):
    """Ranks as host threads on one GPU, each with its own solver handle + native driver over an in-process communicator."""
    from cuba_amd.capi import HipSolver
    from cuba_amd.dist import NativeDist
    comms = ThreadComm.create(world)
    out, err = [None] * world, []

    def work(c):
        try:
            if ranged:
                h = HipSolver(None, RK_HUBER, precision=precision, **options)
                h.set_graph(_values_only_inside(fp, landmark_ranges(fp.eL, fp.Lt, world)[c.rank]), landmark_range=landmark_ranges(fp.eL, fp.Lt, world)[c.rank])
            else:
                h = HipSolver(fp, RK_HUBER, precision=precision, **options)
            d = NativeDist(h, fp, c.rank, world, comm=c, precision=precision)
            chi2 = d.optimize(iters)
            if parts_out is not None:
                parts_out[c.rank] = d.reduction_parts()
            out[c.rank] = (chi2, d.complete_solution(), d.counters())
            d.close()
        except Exception as e:   # pragma: no cover
            err.append(e)
            c.s.barrier.abort()
    th = [threading.Thread(target=work, args=(c,)) for c in comms]
    [t.start() for t in th]; [t.join() for t in th]
    assert not err, err
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_native_driver_emulated_ranks(world):
    """The C++ driver (cuba_hip_dist_optimize) with 2 / 3 ranks emulated as threads on one GPU: same trajectory and
    estimates as the single-handle solve; one large all-reduce per trial (+1 for lambda_0), bit-identical replicas."""
    from cuba_amd.capi import HipSolver
    fp = flatten(synth_ba(120, 6000, 24000, seed=9))
    single = HipSolver(fp, RK_HUBER)
    want = single.optimize(8)["chi2"]
    q1, t1, X1 = single.state()
    out = _run_native_ranks(fp, world, 8, None)
    for chi2, (q, t, X), c in out:
        assert len(chi2) == len(want) and np.all(np.abs(chi2 - want) <= 1e-8 * want)
        assert np.abs(X - X1).max() < 1e-6 and np.abs(t - t1).max() < 1e-6 and np.abs(q - q1).max() < 1e-8
        assert c["large_allreduces"] == c["lm_trials"] + 1 + 1          # per trial + lambda_0 + complete_solution
        assert c["small_allreduces"] == c["lm_trials"] + 1 + 1          # evaluation per trial + first F + max-diagonal
    assert all(np.array_equal(out[0][0], o[0]) for o in out[1:])
    assert all(np.array_equal(a, b) for o in out[1:] for a, b in zip(out[0][1], o[1]))


@pytest.mark.gpu
@pytest.mark.parametrize("device_setup", [1, 0])
def test_schur_in_parts_is_the_schur_pass_bit_for_bit(device_setup):
    """cuba_hip_schur_part (include/cuba_hip.h): a landmark-partitioned handle cuts its reduced matrix at block rows into ranges and runs
    the block pass range by range.  The ranges tile the matrix part of the reduction buffer, part 0 also reports [bsc | bp]; every
    reported range is final when its part has run (the later parts do not touch it), and the whole buffer equals the one-pass result
    of a handle without the cut bit for bit -- for both structure builders, on every rank."""
    import torch
    from cuba_amd.capi import HipSolver
    fp = flatten(synth_ba(120, 6000, 24000, seed=9))
    whole = HipSolver(fp, RK_HUBER)
    lam = 1e-5 * whole.max_diagonal()
    for lo, hi in landmark_ranges(fp.eL, fp.Lt, 3):
        one = HipSolver(fp, RK_HUBER, reduction_chunks=1, device_setup=device_setup)
        one.set_partition(lo, hi)
        assert one.schur_parts() == 1
        one.set_lambda(lam); one.schur()
        want = {k: one.array(k) for k in ("hsc", "bsc", "bp")}
        h = HipSolver(fp, RK_HUBER, reduction_chunks=5, device_setup=device_setup)
        h.set_partition(lo, hi)
        nparts = h.schur_parts()
        assert 2 <= nparts <= 5
        h.set_lambda(lam)
        nhsc = want["hsc"].size
        covered, snaps = 0, []
        for c in range(nparts):
            (o0, n0), (o1, n1) = h.schur_part(c)
            assert o0 == covered and n0 > 0 and n0 % 36 == 0
            covered += n0
            assert (o1, n1) == ((nhsc, want["bsc"].size + want["bp"].size) if c == 0 else (0, 0))
            snaps.append((o0, n0, h.array("hsc")[o0:o0 + n0].copy()))
            if c == 0:
                assert np.array_equal(h.array("bsc"), want["bsc"]) and np.array_equal(h.array("bp"), want["bp"])
        assert covered == nhsc
        got = h.array("hsc")
        assert np.array_equal(got, want["hsc"])
        for o0, n0, part in snaps:
            assert np.array_equal(part, got[o0:o0 + n0])
        # the one-call stage of such a handle runs all parts
        h.set_lambda(2 * lam); h.schur(); one.set_lambda(2 * lam); one.schur()
        assert np.array_equal(h.array("hsc"), one.array("hsc")) and np.array_equal(h.array("bsc"), one.array("bsc"))
        with pytest.raises(Exception):
            h.schur_part(nparts)
    assert whole.schur_parts() == 1


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_native_driver_sums_the_reduction_in_parts(precision):
    """The native driver with the reduction cut into parts: one all-reduce per block-row range, issued behind
    the block pass of that range on a second stream, plus one for [bsc | bp]: same trajectory and estimates as with one all-reduce per
    trial, bit for bit, on every rank; the counters count the parts."""
    fp = flatten(synth_ba(120, 6000, 24000, seed=9))
    world, iters = 3, 6
    base = _run_native_ranks(fp, world, iters, None, precision=precision, reduction_chunks=1)
    parts = [None] * world
    out = _run_native_ranks(fp, world, iters, None, precision=precision, parts_out=parts, reduction_chunks=4)
    nparts = parts[0][0]
    assert 2 <= nparts <= 4 and all(p[0] == nparts for p in parts)
    for (chi2, sol, c), (chi2b, solb, cb) in zip(out, base):
        assert np.array_equal(chi2, chi2b)
        assert all(np.array_equal(a, b) for a, b in zip(sol, solb))
        assert c["lm_trials"] == cb["lm_trials"] and c["small_allreduces"] == cb["small_allreduces"]
        assert c["large_allreduces"] == c["lm_trials"] * (nparts + 1) + 1 + 1          # per trial: ranges + [bsc | bp]; lambda_0; complete_solution
        assert c["large_elements"] == cb["large_elements"]
    for p, (_, _, c) in zip(parts, out):
        assert p[1] == c["lm_trials"] * nparts                                           # all but the last range's sum went out under a later part


@pytest.mark.gpu
def test_rank_upload_sends_only_its_own_values():
    """cuba_hip_set_graph_partition (include/cuba_hip.h): a rank uploads the index arrays whole and the
    measurements / information of its own landmarks' edges only -- here the other edges' values are NaN on the host, so anything that read
    them would show.  Same reduced-system contribution, chi2 and per-edge chi2 as whole upload + cuba_hip_set_partition, bit for bit;
    36 bytes per owned edge cross PCIe; the native driver on such handles walks the trajectory of fully loaded ones."""
    from cuba_amd.capi import HipSolver
    fp = flatten(synth_ba(120, 6000, 24000, seed=9))
    whole = HipSolver(fp, RK_HUBER)
    lam = 1e-5 * whole.max_diagonal()
    world, sent = 3, 0
    for lo, hi in landmark_ranges(fp.eL, fp.Lt, world):
        a = HipSolver(fp, RK_HUBER)
        a.set_partition(lo, hi)
        b = HipSolver(None, RK_HUBER)
        b.set_graph(_values_only_inside(fp, (lo, hi)), landmark_range=(lo, hi))
        own = (fp.eL >= lo) & (fp.eL < hi)
        assert b.counter("value_bytes_uploaded") == 36 * int(own.sum()) and a.counter("value_bytes_uploaded") == 32 * fp.E
        sent += b.counter("value_bytes_uploaded")
        assert a.compute_errors() == b.compute_errors()
        ca, cb = a.chi_squares(), b.chi_squares()
        assert np.array_equal(ca, cb) and np.all(cb[~own] == 0) and np.all(cb[own] > 0)
        for h in (a, b):
            h.set_lambda(lam); h.schur()
        for k in ("hsc", "bsc", "bp"):
            assert np.array_equal(a.array(k), b.array(k)), k
        assert b.solve_reduced() and a.solve_reduced()
        assert np.array_equal(a.array("xp"), b.array("xp"))
        with pytest.raises(Exception):          # the other landmarks' values never reached the device: no wider range without a new upload
            b.set_partition(0, -1)
        b.set_partition(lo, (lo + hi) // 2)     # (a narrower one is fine)
    assert sent == 36 * fp.E
    base = _run_native_ranks(fp, world, 6, None)
    out = _run_native_ranks(fp, world, 6, None, ranged=True)
    for (chi2, sol, c), (chi2b, solb, cb) in zip(out, base):
        assert np.array_equal(chi2, chi2b) and all(np.array_equal(x, y) for x, y in zip(sol, solb)) and c == cb
    # a bad range is refused before the previous graph is touched
    with pytest.raises(Exception):
        whole.set_graph(fp, landmark_range=(5, fp.Lt + 1))
    assert whole.compute_errors() > 0


@pytest.mark.gpu
def test_partition_structure_device_setup_equals_host_setup():
    """A landmark-partitioned handle builds its structure on the GPU like any other handle (global block pattern, sub-ranges of the
    global product / pose-edge lists for its landmarks); the host pipeline (device_setup = 0) builds compacted local lists.  Same
    reduced-system contributions bit for bit, and the contributions of all ranks add up to the whole-graph system."""
    from cuba_amd.capi import HipSolver
    fp = flatten(synth_ba(120, 6000, 24000, seed=9))
    whole = HipSolver(fp, RK_HUBER, pose_reorder=0)
    lam = 1e-5 * whole.max_diagonal()
    whole.set_lambda(lam); whole.schur()
    want = {k: whole.array(k) for k in ("hsc", "bsc", "bp")}
    world = 3
    total = {k: np.zeros_like(v) for k, v in want.items()}
    for r, (lo, hi) in enumerate(landmark_ranges(fp.eL, fp.Lt, world)):
        got = []
        for opts in (dict(), dict(device_setup=0)):
            h = HipSolver(fp, RK_HUBER, pose_reorder=0, **opts)
            h.set_partition(lo, hi)
            h.set_lambda(lam); h.schur()
            got.append({k: h.array(k) for k in want})
            assert h.compute_errors() > 0
        for k in want:
            assert np.array_equal(got[0][k], got[1][k]), (r, k)
            total[k] += got[0][k]
    for k in want:
        assert np.abs(total[k] - want[k]).max() <= 1e-11 * np.abs(want[k]).max(), k


@pytest.mark.gpu
def test_native_driver_shuffled_pose_ids_keeps_the_internal_pose_order():
    """Arbitrary vertex ids in landmark-partitioned mode: the structure is built on the device, so the internal trajectory order of
    the poses (which the two-level preconditioner needs) is kept -- round 2 fell back to the caller's order there and needed 20 x the
    PCG iterations.  Same trajectory as the single handle, iteration counts of the id-ordered graph, replicas bit-identical."""
    import copy
    from cuba_amd.capi import HipSolver
    from cuba_amd.dist import NativeDist
    from cuba_amd.synth import synth_named
    g = synth_named("kitti07")
    rng = np.random.default_rng(5)
    perm = rng.permutation(g.nposes); perm[perm == 0], perm[0] = perm[0], 0
    h = copy.deepcopy(g)
    lut = np.zeros(int(g.pose_ids.max()) + 1, dtype=np.int64); lut[g.pose_ids] = perm
    h.pose_ids = perm.astype(np.int64); h.mono_vp = lut[g.mono_vp]; h.stereo_vp = lut[g.stereo_vp]
    fp, fp_ord = flatten(h), flatten(g)
    ordered = HipSolver(fp_ord, RK_HUBER); ordered.optimize(6)
    it_ord = int(ordered.pcg_history()[0].sum())
    single = HipSolver(fp, RK_HUBER); want = single.optimize(6)["chi2"]
    world = 2
    comms = ThreadComm.create(world)
    out, err = [None] * world, []

    def work(c):
        try:
            hh = HipSolver(fp, RK_HUBER)
            d = NativeDist(hh, fp, c.rank, world, comm=c)
            chi2 = d.optimize(6)
            out[c.rank] = (chi2, int(np.abs(hh.pcg_history()[0]).sum()), hh.pcg_history()[1])
            d.close()
        except Exception as e:   # pragma: no cover
            err.append(e)
            c.s.barrier.abort()
    th = [threading.Thread(target=work, args=(c,)) for c in comms]
    [t.start() for t in th]; [t.join() for t in th]
    assert not err, err
    for chi2, its, bad in out:
        assert len(chi2) == len(want) and np.all(np.abs(chi2 - want) <= 1e-8 * want)
        assert bad == 0 and its <= 1.15 * it_ord, (its, it_ord)
    assert np.array_equal(out[0][0], out[1][0])


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_native_driver_float32_build_emulated_ranks(world):
    """BASELINE configs[4] "plus USE_FLOAT32 variant" (ref option: src/scalar.h:25-29, CMakeLists.txt:7): libcuba_hip_dist_f32.so
    driving libcuba_hip_f32.so handles, 2 and 8 ranks emulated as threads.  Replicas bit-identical (fixed summation orders in
    fp32 as in fp64), the trajectory follows the fp32 single handle to fp32 resolution and the oracle to the stated fp32
    tolerance (chi2 1e-4 relative)."""
    from cuba_amd.capi import HipSolver
    from oracle.oracle import OracleSolver
    fp = flatten(synth_ba(120, 6000, 24000, seed=9))
    ref = OracleSolver(fp, RK_HUBER).optimize(8)["chi2"]
    single = HipSolver(fp, RK_HUBER, precision="f32")
    assert single.scalar_size == 4
    want = single.optimize(8)["chi2"]
    out = _run_native_ranks(fp, world, 8, None, precision="f32")
    for chi2, (q, t, X), c in out:
        m = min(len(chi2), len(ref))
        assert m >= 6                                                       # fp32 may stop early once the gain is below its resolution
        assert np.all(np.abs(chi2[:m] - ref[:m]) <= 1e-4 * ref[:m])
        mm = min(len(chi2), len(want))
        assert np.all(np.abs(chi2[:mm] - want[:mm]) <= 1e-4 * want[:mm])     # partial sums are added in another order than on one handle
        assert c["large_allreduces"] == c["lm_trials"] + 1 + 1
    assert all(np.array_equal(out[0][0], o[0]) for o in out[1:])
    assert all(np.array_equal(a, b) for o in out[1:] for a, b in zip(out[0][1], o[1]))


@pytest.mark.gpu
def test_failed_driver_creation_leaves_the_solver_handle_unrestricted():
    """cuba_hip_dist_create_custom with an incomplete collective table fails AFTER it has bound the solver handle; the handle must
    come back with its full landmark range (round-2 advisor finding) -- a single-GPU optimize on it equals a fresh handle's."""
    import ctypes as C
    from cuba_amd.capi import HipSolver
    from cuba_amd.dist import load_dist_library
    fp = flatten(synth_ba(**GRAPH))
    want = HipSolver(fp, RK_HUBER).optimize(5)["chi2"]
    h = HipSolver(fp, RK_HUBER)
    lib = load_dist_library()
    FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)

    class Ops(C.Structure):
        _fields_ = [("ctx", C.c_void_p), ("allreduce_sum", FN), ("allreduce_max", FN)]
    d = C.c_void_p()
    # (a) rejected before binding: incomplete table
    assert lib.cuba_hip_dist_create_custom(h.h, C.byref(Ops(None, FN(0), FN(0))), 0, 2, 0, fp.Lt // 2, C.byref(d)) != 0 and not d.value
    # (b) fails after set_partition + build_structure (injected: what a refused ncclCommInitRank looks like)
    import os
    os.environ["CUBA_HIP_DIST_TEST_FAIL_AFTER_BIND"] = "1"
    try:
        ok = lambda *a: 0      # noqa: E731
        ops = Ops(None, FN(ok), FN(ok))
        assert lib.cuba_hip_dist_create_custom(h.h, C.byref(ops), 0, 2, 0, fp.Lt // 2, C.byref(d)) != 0 and not d.value
    finally:
        del os.environ["CUBA_HIP_DIST_TEST_FAIL_AFTER_BIND"]
    got = h.optimize(5)["chi2"]
    assert np.array_equal(got, want)
    h.set_partition(0, fp.Lt // 2); h.set_partition(0, -1)          # and the explicit way to lift a restriction
    h.set_state(fp.q, fp.t, fp.Xw)
    assert np.array_equal(h.optimize(5)["chi2"], want)


@pytest.mark.gpu
def test_native_driver_over_a_real_rccl_communicator_single_rank():
    """A 1-rank RCCL communicator (all this box can host: RCCL wants one device per rank): ncclCommInitRank + the whole
    native loop on the solver's stream; must reproduce cuba_hip_optimize -- bit for bit against the same stage kernels
    (the host loop: "profile" = 1), to summation-order noise against the default fused tail (which adds the chi2 partials per landmark
    workgroup instead of per edge stride)."""
    from cuba_amd.capi import HipSolver
    from cuba_amd.dist import NativeDist, rccl_unique_id
    fp = flatten(synth_ba(120, 6000, 24000, seed=9))
    want = HipSolver(fp, RK_HUBER, profile=1).optimize(6)["chi2"]
    want_fused = HipSolver(fp, RK_HUBER).optimize(6)["chi2"]
    h = HipSolver(fp, RK_HUBER)
    d = NativeDist(h, fp, 0, 1, unique_id=rccl_unique_id())
    got = d.optimize(6)
    assert np.array_equal(got, want)
    assert len(got) == len(want_fused) and np.all(np.abs(got - want_fused) <= 1e-12 * want_fused)
    d.close()


@pytest.mark.gpu
def test_reduction_in_parts_over_a_real_rccl_communicator_single_rank(monkeypatch):
    """The stream / event choreography of the reduction in parts over the real library, as far as one GPU allows: a 1-rank RCCL
    communicator whose collectives are issued anyway (CUBA_HIP_DIST_SINGLE_RANK_COLLECTIVES), ncclAllReduce of every block-row range on the
    driver's second stream behind an event of the solver's stream.  Same trajectory as one all-reduce per trial, bit for bit."""
    from cuba_amd.capi import HipSolver
    from cuba_amd.dist import NativeDist, rccl_unique_id
    fp = flatten(synth_ba(120, 6000, 24000, seed=9))
    monkeypatch.setenv("CUBA_HIP_DIST_SINGLE_RANK_COLLECTIVES", "1")
    runs = []
    for chunks in (1, 4):
        h = HipSolver(fp, RK_HUBER, reduction_chunks=chunks)
        d = NativeDist(h, fp, 0, 1, unique_id=rccl_unique_id())
        chi2 = d.optimize(6)
        runs.append((chi2, h.state(), d.counters(), d.reduction_parts()))
        d.close()
    (c1, s1, k1, p1), (c4, s4, k4, p4) = runs
    assert p1 == (1, 0) and 2 <= p4[0] <= 4 and p4[1] == k4["lm_trials"] * p4[0]
    assert np.array_equal(c1, c4) and all(np.array_equal(a, b) for a, b in zip(s1, s4))
    assert k4["large_allreduces"] == k4["lm_trials"] * (p4[0] + 1) + 1 and k1["large_allreduces"] == k1["lm_trials"] + 1


def _gpu_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:   # pragma: no cover
        return 0


@pytest.mark.gpu
@pytest.mark.skipif(_gpu_count() < 2, reason="a multi-rank RCCL communicator needs one device per rank (this box has fewer than 2)")
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_native_driver_over_rccl_two_ranks(tmp_path, precision):
    """The first thing to run on a multi-GPU lease: two processes, one GPU each, the native driver over a REAL 2-rank RCCL
    communicator (cuba_hip_dist_create_rccl: ncclCommInitRank from a unique id, in-stream ncclAllReduce of [Hsc | bsc | bp] and
    of the two evaluation scalars).  Must reproduce the single-handle solve (chi2 1e-8, estimates 1e-6; fp32 library: the fp32
    bars), both ranks bit-identical, exactly trials + 1 large all-reduces."""
    import torch.multiprocessing as mp
    from dist_helpers import rccl_rank_main
    from cuba_amd.capi import HipSolver
    graph = dict(P=120, L=6000, E=24000, seed=9)
    iters = 6
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(rccl_rank_main, args=(2, port, str(tmp_path), graph, RK_HUBER, iters, precision), nprocs=2, join=True)
    r = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(2)]
    fp = flatten(synth_ba(**graph))
    single = HipSolver(fp, RK_HUBER, precision=precision)
    want = single.optimize(iters)["chi2"]
    q1, t1, X1 = single.state()
    ctol, etol = (1e-8, 1e-6) if precision == "f64" else (1e-4, 1e-3)
    for k in range(2):
        chi2 = np.array(r[k]["chi2"])
        assert len(chi2) == len(want) and np.all(np.abs(chi2 - want) <= ctol * want), (k, chi2, want)
        assert np.abs(np.array(r[k]["X"]) - X1).max() < etol and np.abs(np.array(r[k]["t"]) - t1).max() < etol
        assert r[k]["counters"]["large_allreduces"] == iters + 1 and r[k]["counters"]["lm_trials"] == iters
    assert r[0]["chi2"] == r[1]["chi2"] and r[0]["q"] == r[1]["q"] and r[0]["t"] == r[1]["t"]     # replicated solve: bit-identical ranks


@pytest.mark.gpu
@pytest.mark.skipif(_gpu_count() < 2, reason="needs 2 GPUs (the driver's scaling run, RCCL backend)")
def test_bench_two_gpus_over_rccl():
    """bench.py --gpus 2 exactly as the driver launches it on a multi-GPU node (RCCL): every rank's graph checked against its
    committed golden trajectory inside the line (config 4), and config 5's landmark-partitioned leg over a real 2-rank
    communicator in the same invocation."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("CUBA_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "10"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["config"]["graphs"] == 2
    par = rec["rank_parity"]
    assert len(par) == 2 and all(p["chi2_max_rel_diff_vs_golden"] <= 1e-6 for p in par), par
    part = rec["partitioned"]
    assert "error" not in part, part
    assert part["iterations_done"] == 10 and part["chi2_max_rel_diff_vs_golden"] <= 1e-6


@pytest.mark.gpu
def test_bench_partition_two_ranks_share_one_gpu_over_gloo():
    """bench.py --gpus 2 --partition launched the way the driver launches it (torch.distributed.run, 127.0.0.1), with the
    gloo backend because RCCL wants one device per rank: TorchComm + device views + the native driver's custom-collective
    path execute end to end, and the line must be a valid strong-scaling record."""
    import os
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, CUBA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--partition", "--shape", "kitti07",
           "--steps", "10", "--warmup", "10", "--no-cpu-baseline", "--no-end-to-end"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["value"] > 0 and rec["lm_trials"] == 10
    # same physical problem as the single-GPU run of this shape: before its timed part (whose runs start from perturbed estimates)
    # the partitioned handle ran 10 iterations from the generator's initial guess against the committed oracle trajectory ...
    par = rec["rank_parity"]
    assert len(par) == 2 and all(p["seed"] == 7 and p["chi2_max_rel_diff_vs_golden"] <= 1e-6 for p in par), par
    # ... which is the live oracle's (the CPU suite re-derives that golden entry too)
    from oracle.oracle import OracleSolver
    from cuba_amd.synth import synth_named
    ref = OracleSolver(flatten(synth_named("kitti07")), RK_HUBER).optimize(10)["chi2"]
    assert abs(par[0]["chi2_last"] - ref[-1]) <= 1e-6 * ref[-1]


@pytest.mark.gpu
def test_bench_independent_graphs_two_ranks_share_one_gpu_over_gloo():
    """bench.py --gpus 2 exactly as the driver launches it for the scaling runs (one independent KITTI-00-sized graph per rank,
    no data-path collective, weak scaling); gloo instead of RCCL because both ranks share this box's one GPU."""
    import os
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, CUBA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "10"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["config"]["graphs"] == 2
    assert rec["lm_trials"] == 20 and rec["value"] > 0 and rec["steps"] == 20 and rec["warmup"] == 10
    assert "roofline" in rec and rec["roofline"]["path"]["trials"] == 20
    # every rank compared its own graph (seed 100 + rank) with the committed oracle trajectory before the timed region
    par = rec["rank_parity"]
    assert [p["seed"] for p in par] == [100, 101] and all(p["chi2_max_rel_diff_vs_golden"] <= 1e-6 for p in par), par
    # the same line also carries config 5's mode (one graph, landmark-partitioned, native driver) measured in the same invocation
    part = rec["partitioned"]
    assert "error" not in part, part
    assert part["scaling"] == "strong" and part["iterations_done"] == 10 and part["value"] > 0
    # ... with where its time goes: the replicated reduced solve bounds the speed-up of this mode (Amdahl)
    ts = part["time_shares"]
    assert "error" not in ts, ts
    assert 0 < ts["replicated_reduced_solve"] < 1 and ts["amdahl_ceiling_speedup"] > 0 and part["speedup_over_one_gpu"] > 0


@pytest.mark.gpu
def test_degenerate_inputs_do_not_crash():
    """Empty edge list, a single free pose, a single landmark, a rank upload with an empty / the whole range: every entry point returns
    (status or result), nothing hangs."""
    from cuba_amd.capi import CubaHipError, HipSolver
    from cuba_amd.graph import FlatProblem
    g = synth_ba(40, 600, 2400, seed=1)
    fp = flatten(g)
    # no edges at all
    empty = FlatProblem(Pt=2, Pf=1, Lt=1, Lf=1, q=fp.q[:2].copy(), t=fp.t[:2].copy(), cam=fp.cam[:2].copy(), Xw=fp.Xw[:1].copy(),
                        eP=np.zeros(0, np.int32), eL=np.zeros(0, np.int32), eDim=np.zeros(0, np.uint8), meas=np.zeros((0, 3)), omega=np.zeros(0),
                        pose_src=np.arange(2), lm_src=np.arange(1), edge_src=np.zeros(0, np.int64))
    h = HipSolver(empty, RK_HUBER)
    assert h.compute_errors() == 0.0
    try:
        chi2 = h.optimize(3)["chi2"]
        assert len(chi2) <= 3 and np.all(np.isfinite(chi2))
    except CubaHipError:
        pass                                                        # a reported failure is fine, a crash or a hang is not
    # one free pose (+ one fixed), all landmarks free
    keep = (fp.eP == 0) | (fp.eP == fp.Pf)                           # edges of free pose 0 and of the fixed pose
    lm = np.unique(fp.eL[keep]); lut = -np.ones(fp.Lt, np.int64); lut[lm] = np.arange(len(lm))
    one = FlatProblem(Pt=2, Pf=1, Lt=len(lm), Lf=len(lm), q=fp.q[[0, fp.Pf]].copy(), t=fp.t[[0, fp.Pf]].copy(), cam=fp.cam[[0, fp.Pf]].copy(),
                      Xw=fp.Xw[lm].copy(), eP=np.where(fp.eP[keep] == 0, 0, 1).astype(np.int32), eL=lut[fp.eL[keep]].astype(np.int32),
                      eDim=fp.eDim[keep].copy(), meas=fp.meas[keep].copy(), omega=fp.omega[keep].copy(),
                      pose_src=np.arange(2), lm_src=np.arange(len(lm)), edge_src=np.arange(int(keep.sum())))
    from oracle.oracle import OracleSolver
    ref = OracleSolver(one, RK_HUBER).optimize(4)["chi2"]
    got = HipSolver(one, RK_HUBER).optimize(4)["chi2"]
    assert len(got) == len(ref) and np.all(np.abs(got - ref) <= 1e-6 * ref)
    # a rank upload with an empty range, with no edges, and more reduction ranges asked for than the matrix has block rows
    h = HipSolver(None, RK_HUBER, reduction_chunks=64)
    h.set_graph(fp, landmark_range=(7, 7))
    assert h.counter("value_bytes_uploaded") == 0 and h.compute_errors() == 0.0
    h.set_lambda(1.0); h.schur()
    assert 1 <= h.schur_parts() <= 64
    h.set_graph(fp, landmark_range=(0, fp.Lt))                           # the whole range: every value goes up, 36 bytes each
    assert h.counter("value_bytes_uploaded") == 36 * fp.E
    want = HipSolver(fp, RK_HUBER)
    assert h.compute_errors() == want.compute_errors()
    parts = h.schur_parts()
    lam = 1e-5 * want.max_diagonal()
    h.set_lambda(lam); want.set_lambda(lam); h.schur(); want.schur()
    assert parts >= 2 and np.array_equal(h.array("hsc"), want.array("hsc")) and np.array_equal(h.array("bsc"), want.array("bsc"))
    e = HipSolver(None, RK_HUBER)
    e.set_graph(empty, landmark_range=(0, 1))
    assert e.compute_errors() == 0.0 and e.schur_parts() == 1
