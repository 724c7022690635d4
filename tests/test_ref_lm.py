"""LM-level pin against the REFERENCE ITSELF: the reference's own optimiser -- CudaBundleAdjustmentImpl::optimize,
CudaBlockSolver and all kernels, compiled from /root/reference/src in place (oracle/ref_build, only the closed-source
cuSOLVER step is a stand-in exact Cholesky) -- runs whole Levenberg-Marquardt trajectories on the MI355X.  The CPU oracle
and the HIP path must follow them: chi2 per iteration, number of executed iterations, final estimates, per-edge chi2,
including the degenerate modes that go through the reference's gpu::solveDiagonalSystem (pose-only, landmark-only) and a
start that makes the reference reject trials.  Skipped when oracle/_ref was not built (needs the reference checkout).

What is NOT here, and why: a BASELINE-size run with the Huber kernel AND rejected trials.  With Huber the reference only starts rejecting at
lambda ~ 1e-14 x max-diag (iteration ~25 of a 30-iteration run), where the run is chaotic -- the oracle with 1 and with 16 threads ends 40 %
apart -- so no implementation can be pinned to another there; rejections at BASELINE size are pinned without a robust kernel
(k00_rot0.5rad_none) and with Tukey (k00_lm10m_tukey, reference-vs-itself spread as the yardstick), Huber with rejections at <= 120 poses."""
import copy

import numpy as np
import pytest

from conftest import RK_HUBER, RK_NONE, RK_TUKEY, with_fixed
from cuba_amd.graph import flatten, write_back
from cuba_amd.synth import synth_ba
from oracle import ref_lm

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref_lm.available(), reason="oracle/_ref/libcuba_ref_lm.so not built")]

ORACLE_TOL = 1e-9     # same algorithm, exact solves on both sides: summation-order noise only (the reference sums with atomics)
HIP_TOL = 1e-8        # HIP path at pcg_tol = 1e-11; at the default pcg_tol the bar is the north star's 1e-6


def rough_start(g, seed=1, sx=3.0, st=0.6, sr=0.0):
    """landmarks + N(0, sx) m, free-pose translations + N(0, st) m, free-pose rotations composed with exp(N(0, sr) rad)"""
    h = copy.deepcopy(g)
    rng = np.random.default_rng(seed)
    h.lm_X = h.lm_X + rng.normal(0, sx, h.lm_X.shape)
    free = ~h.pose_fixed
    h.pose_t = h.pose_t.copy(); h.pose_t[free] += rng.normal(0, st, (int(free.sum()), 3))
    if sr > 0:
        from scipy.spatial.transform import Rotation
        dq = Rotation.from_rotvec(rng.normal(0, sr, (int(free.sum()), 3)))
        h.pose_q = h.pose_q.copy(); h.pose_q[free] = (dq * Rotation.from_quat(h.pose_q[free])).as_quat()
    return h


def cases():
    g = synth_ba(40, 600, 2400, seed=1)
    yield "huber", g, RK_HUBER, 10
    yield "none", g, RK_NONE, 8
    yield "fixed_vertices", with_fixed(g, fixed_pose_rows=[3, 4, 20], fixed_lm_rows=list(range(0, 300, 7))), RK_HUBER, 8
    yield "pose_only", with_fixed(g, fixed_lm_rows=range(g.nlandmarks)), RK_HUBER, 6            # ref: solveDiagonalSystem(Hpp)
    yield "landmark_only", with_fixed(g, fixed_pose_rows=range(g.nposes)), RK_HUBER, 6           # ref: solveDiagonalSystem(Hll)
    yield "tukey_rejected_trials", rough_start(synth_ba(60, 1500, 6000, seed=3)), RK_TUKEY, 12
    yield "kitti07_like", synth_ba(120, 6000, 24000, seed=9), RK_HUBER, 10


def full_size_cases():
    """BASELINE configs[0] and configs[1] at full size (round-2 verdict: the reference-run pins stopped at 120 poses).
    KITTI-07 shape: n = 1482, the reference's stand-in linear solver factorises it on the host as before; KITTI-00 shape:
    n = 7986, the same exact dense Cholesky through rocSOLVER on the device (oracle/ref_build/ref_linear_solver.cpp).
    Round 4: S2M (configs[2]; n = 29 994 -- 7.2 GB dense, ~9 TFLOP per factorisation, 11 of them) and configs[4]'s graph
    (n = 59 994 -- 28.8 GB dense, ~72 TFLOP per factorisation, ~80 s; run by default since round 6, CUBA_TEST_SKIP_REF_G4M=1 opts out).
    name -> (graph factory, robust kernels, iterations): graphs are generated only for the case that runs."""
    from cuba_amd.synth import synth_named
    return {"kitti07_full": (lambda: synth_named("kitti07"), RK_HUBER, 10),
            "kitti00_full": (lambda: synth_named("kitti00"), RK_HUBER, 10),
            "s2m_full": (lambda: synth_named("s2m"), RK_HUBER, 10),
            "g4m_full": (lambda: synth_named("g4m"), RK_HUBER, 10)}


def rejected_trial_cases():
    """Full KITTI-00 shape from starts that make the REFERENCE reject trials (restore + lambda *= nu, nu *= 2:
    src/cuda_bundle_adjustment.cpp:824-845) inside a well-conditioned stretch of the run.  Found with the oracle: rejections in
    iterations 10 and 12 / 11 and 13 (three trials each), and the trajectory is stable against summation-order noise (1 vs 16
    oracle threads: <= 5e-12 / 1e-9 on chi2).  A 30-iteration Huber run that only starts rejecting at lambda ~ 1e-14 x max-diag is
    chaotic instead -- 40 % apart between 1 and 16 threads -- and cannot pin anything."""
    from cuba_amd.synth import synth_named
    return {"k00_rot0.5rad_none": (lambda: rough_start(synth_named("kitti00"), seed=1, sx=0.0, st=0.0, sr=0.5), RK_NONE, 14),
            "k00_lm10m_tukey": (lambda: rough_start(synth_named("kitti00"), seed=1, sx=10.0, st=0.0, sr=0.0), RK_TUKEY, 14),
            # round 6: the same hard start at S2M size (4999 free poses, n = 29 994): the exact solver is sparse now and has no size limit
            "s2m_lm10m_tukey": (lambda: rough_start(synth_named("s2m"), seed=1, sx=10.0, st=0.0, sr=0.0), RK_TUKEY, 14)}


def in_graph_order(fp, g, state):
    """solver-order (q, t, X) -> the graph's row order (vertices the solver left out keep the graph's values)"""
    h = copy.deepcopy(g)
    write_back(h, fp, *state)
    return h.pose_q, h.pose_t, h.lm_X


@pytest.mark.parametrize("name,g,rk,iters", list(cases()), ids=[c[0] for c in cases()])
def test_lm_trajectory_follows_the_reference(name, g, rk, iters):
    from cuba_amd.capi import HipSolver
    from oracle.oracle import OracleSolver
    ref = ref_lm.run(g, rk, iters)
    fp = flatten(g)
    o = OracleSolver(fp, rk); ro = o.optimize(iters)
    assert len(ro["chi2"]) == len(ref["chi2"]), (len(ro["chi2"]), len(ref["chi2"]))
    # (the non-convex Tukey run from a rough start amplifies summation-order noise along its rejected / re-tried steps:
    # measured 2.3e-9 there, <= 1e-10 everywhere else)
    otol = 1e-7 if name == "tukey_rejected_trials" else ORACLE_TOL
    assert np.all(np.abs(ro["chi2"] - ref["chi2"]) <= otol * ref["chi2"]), np.abs(ro["chi2"] / ref["chi2"] - 1).max()
    if name == "tukey_rejected_trials":
        assert ro["trials"].max() > 1                                  # the reference rejected the same trials
    h = HipSolver(fp, rk, pcg_tol=1e-11); rh = h.optimize(iters)["chi2"]
    htol = 1e-7 if name == "tukey_rejected_trials" else HIP_TOL
    assert len(rh) == len(ref["chi2"]) and np.all(np.abs(rh - ref["chi2"]) <= htol * ref["chi2"]), np.abs(rh / ref["chi2"] - 1).max()
    hd = HipSolver(fp, rk); rd = hd.optimize(iters)["chi2"]             # default tolerance: the stated 1e-6
    # (the rough Tukey run amplifies the 1e-7 solve tolerance along its rejected / re-tried steps: measured 2.8e-6)
    assert len(rd) == len(ref["chi2"]) and np.all(np.abs(rd - ref["chi2"]) <= (1e-5 if name == "tukey_rejected_trials" else 1e-6) * ref["chi2"])
    # final estimates (the reference wrote them back into its vertex objects, finalize() :512-526)
    # (Tukey gives outlier-only landmarks zero weight: their position is held by the damping term alone and moves by
    # 1e-5 m for last-bit differences in the sums -- measured 2.9e-5 m between the reference and the oracle there)
    rough = name == "tukey_rejected_trials"
    for label, solver, lim in (("oracle", o, 1e-3 if rough else 1e-7), ("hip", h, 1e-3 if rough else 1e-6)):
        for a, b in zip(in_graph_order(fp, g, solver.state()), (ref["q"], ref["t"], ref["Xw"])):
            assert np.abs(a - b).max() <= lim, (label, np.abs(a - b).max())
    # per-edge chi2 at the final estimate, caller's edge order; edges with both ends fixed report 0 on both sides
    per_edge = np.zeros(g.nedges); per_edge[fp.edge_src] = h.chi_squares()
    want = np.concatenate([ref["chi_mono"], ref["chi_stereo"]])
    assert np.abs(per_edge - want).max() <= (1e-3 if rough else 1e-6) * max(1.0, np.abs(want).max())


def test_warm_start_protocol_follows_the_reference():
    """The samples' protocol (sample_comparison_with_g2o.cpp:303-307, 74-79): initialize + optimize(1), then initialize +
    optimize(10) from the written-back estimates."""
    from cuba_amd.capi import HipSolver
    g = synth_ba(40, 600, 2400, seed=1)
    ref = ref_lm.run(g, RK_HUBER, 1, nruns=1)
    g1 = copy.deepcopy(g)
    g1.pose_q, g1.pose_t, g1.lm_X = ref["q"], ref["t"], ref["Xw"]
    ref2 = ref_lm.run(g1, RK_HUBER, 10)
    fp = flatten(g); h = HipSolver(fp, RK_HUBER, pcg_tol=1e-11); h.optimize(1)
    g2 = copy.deepcopy(g); write_back(g2, fp, *h.state())
    fp2 = flatten(g2); h2 = HipSolver(fp2, RK_HUBER, pcg_tol=1e-11)
    got = h2.optimize(10)["chi2"]
    assert len(got) == len(ref2["chi2"]) and np.all(np.abs(got - ref2["chi2"]) <= HIP_TOL * ref2["chi2"])


@pytest.mark.parametrize("name", ["kitti07_full", "kitti00_full", "s2m_full"])
def test_reference_stage_times_on_this_gpu(name):
    """The reference's own kernels, compiled for gfx950 as they are (CUDA -> HIP name shim), timed by the reference's own stage timers
    (CudaBundleAdjustment::timeProfile, src/cuda_bundle_adjustment.cpp:545-562) on this GPU, beside the HIP path's stage timers of the same
    names (option "profile": every stage synchronises, as the reference's do) under the samples' protocol: warm-up initialize() + optimize(),
    then initialize() + optimize(10) from the written-back estimates.  Comparable: "2: Compute Error", "3: Build System", "4: Schur
    Complement", "7: Update Solution" (the reference's kernels against this library's) and "1: Build Structure".  NOT comparable: the
    decomposition stages -- upstream they are cuSOLVER's sparse Cholesky, here a dense rocSOLVER stand-in (oracle/ref_build).  A record
    (printed, and written to gpurun_out/ when that exists), with one claim asserted: the four kernel stages together are faster here."""
    import os
    import time
    from cuba_amd.capi import HipSolver
    make, rk, iters = full_size_cases()[name]
    g = make()
    ref = ref_lm.run(g, rk, iters, nruns=2)                      # stage timers and walls of the SECOND initialize() + optimize()
    fp0 = flatten(g)
    h0 = HipSolver(fp0, rk); h0.optimize(iters)
    g1 = copy.deepcopy(g); write_back(g1, fp0, *h0.state())
    fp1 = flatten(g1)
    walls = {}
    for label, opts in (("profiled", dict(profile=1)), ("plain", dict())):
        h = HipSolver(fp0, rk, **opts); h.optimize(iters)       # warm-up of this handle (structure, graphs)
        t0 = time.perf_counter(); h.set_graph(fp1); t1 = time.perf_counter(); h.optimize(iters); t2 = time.perf_counter()
        walls[label] = (t1 - t0, t2 - t1)
        if label == "profiled":
            prof = h.profile()
    kernel_stages = ("2: Compute Error", "3: Build System", "4: Schur Complement", "7: Update Solution")
    lines = [f"[{name}] stage seconds of initialize() + optimize({iters}) after a warm-up run, this GPU: the reference's kernels (gfx950 build of its CUDA sources) | this library",
             *(f"  {k:28s} {ref['profile'][k] * 1e3:10.3f} ms | {prof[k] * 1e3:9.3f} ms" + ("   (not comparable: dense stand-in for cuSOLVER | PCG)" if k[0] in "56" else "")
               for k in ref_lm.PROFILE_KEYS),
             f"  kernel stages 2 + 3 + 4 + 7      {sum(ref['profile'][k] for k in kernel_stages) * 1e3:10.3f} ms | {sum(prof[k] for k in kernel_stages) * 1e3:9.3f} ms",
             f"  host wall initialize / optimize  {ref['wall_initialize'] * 1e3:.3f} / {ref['wall_optimize'] * 1e3:.3f} ms | profiled {walls['profiled'][0] * 1e3:.3f} / {walls['profiled'][1] * 1e3:.3f} ms, "
             f"plain {walls['plain'][0] * 1e3:.3f} / {walls['plain'][1] * 1e3:.3f} ms"]
    print("\n" + "\n".join(lines))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, f"ref_stage_times_{name}.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    assert all(v >= 0 for v in ref["profile"].values()) and sum(ref["profile"][k] for k in kernel_stages) > 0
    assert sum(prof[k] for k in kernel_stages) < sum(ref["profile"][k] for k in kernel_stages)


@pytest.mark.parametrize("name", ["kitti07_full", "kitti00_full", "s2m_full", "g4m_full"])
def test_full_size_lm_trajectory_follows_the_reference(name):
    """The reference's own optimiser (its LM loop, block solver and all kernels, compiled in place) at the full BASELINE
    shapes, under the samples' protocol (sample_comparison_with_g2o.cpp:303-307): initialize() + optimize(1) warm-up, then
    initialize() + optimize(10) from the written-back estimates.  The CPU oracle must follow it to <= 1e-9 on every
    per-iteration chi2, the HIP path to <= 1e-8 at pcg_tol = 1e-11 and to <= 1e-6 (the north star's bar) at the default
    tolerance; final estimates and per-edge chi2 likewise.  All deviations are printed before anything is asserted."""
    import os
    from cuba_amd.capi import HipSolver
    from oracle.oracle import OracleSolver
    if name == "g4m_full" and os.environ.get("CUBA_TEST_SKIP_REF_G4M") == "1":
        pytest.skip("28.8 GB dense stand-in factorisation, ~72 TFLOP each, ~80 s: skipped on request (CUBA_TEST_SKIP_REF_G4M=1, for small boxes)")
    make, rk, iters = full_size_cases()[name]
    g = make()
    # warm-up run of the reference, then the timed-protocol run from its written-back estimates
    warm = ref_lm.run(g, rk, 1)
    g1 = copy.deepcopy(g); g1.pose_q, g1.pose_t, g1.lm_X = warm["q"], warm["t"], warm["Xw"]
    ref = ref_lm.run(g1, rk, iters)
    assert len(ref["chi2"]) == iters
    fp0 = flatten(g)
    # every implementation runs its OWN warm-up iteration (that is the protocol), then 10 iterations from its own state
    dev = {}
    o0 = OracleSolver(fp0, rk); o0.optimize(1)
    go = copy.deepcopy(g); write_back(go, fp0, *o0.state())
    fpo = flatten(go); o = OracleSolver(fpo, rk); ro = o.optimize(iters)["chi2"]
    dev["oracle chi2"] = float(np.abs(ro / ref["chi2"] - 1).max()) if len(ro) == iters else np.inf
    runs = {}
    for label, opts in (("hip tight", dict(pcg_tol=1e-11)), ("hip default", dict())):
        h0 = HipSolver(fp0, rk, **opts); h0.optimize(1)
        gh = copy.deepcopy(g); write_back(gh, fp0, *h0.state())
        fph = flatten(gh); h = HipSolver(fph, rk, **opts); rh = h.optimize(iters)["chi2"]
        dev[label + " chi2"] = float(np.abs(rh / ref["chi2"] - 1).max()) if len(rh) == iters else np.inf
        runs[label] = (fph, h)
        assert h.pcg_history()[1] == 0
    est = {}
    for label, fp, solver in (("oracle", fpo, o), ("hip tight", *runs["hip tight"]), ("hip default", *runs["hip default"])):
        for nm, a, b in zip("qtX", in_graph_order(fp, g, solver.state()), (ref["q"], ref["t"], ref["Xw"])):
            est[f"{label} {nm}"] = float(np.abs(a - b).max())
    fph, h = runs["hip tight"]
    per_edge = np.zeros(g.nedges); per_edge[fph.edge_src] = h.chi_squares()
    want = np.concatenate([ref["chi_mono"], ref["chi_stereo"]])
    dev["hip tight per-edge chi2 (abs / max)"] = float(np.abs(per_edge - want).max() / max(1.0, np.abs(want).max()))
    print(f"\n[{name}] vs the reference's own optimiser: " + ", ".join(f"{k} {v:.2e}" for k, v in {**dev, **est}.items()))
    assert dev["oracle chi2"] <= 1e-9 and dev["hip tight chi2"] <= 1e-8 and dev["hip default chi2"] <= 1e-6, dev
    for k, v in est.items():
        lim = 1e-7 if k.startswith("oracle") else 1e-6 if "tight" in k else (1e-8 if k.endswith(" q") else 1e-6) * 10
        assert v <= lim, (k, v, lim)
    assert dev["hip tight per-edge chi2 (abs / max)"] <= 1e-6


@pytest.mark.parametrize("name", ["k00_rot0.5rad_none", "k00_lm10m_tukey", "s2m_lm10m_tukey"])
def test_full_size_rejected_trials_follow_the_reference(name):
    """A run at BASELINE size (KITTI-00 shape) in which the reference's own optimiser REJECTS trials -- the restore / lambda *= nu,
    nu *= 2 path of src/cuda_bundle_adjustment.cpp:824-845, so far exercised at <= 60 poses only.  The reference (its LM loop, block
    solver and kernels compiled in place, exact dense Cholesky through rocSOLVER in cuSOLVER's seat) runs 14 iterations from a rough
    start; the oracle must follow it to <= 1e-9 on every per-iteration chi2 with rejected trials in its record, the HIP path to
    <= 1e-8 at pcg_tol = 1e-11 and <= 1e-6 at the default tolerance, with exactly the oracle's number of trials; final estimates
    likewise.
    The Tukey case: with 10 m of landmark noise most observations of many poses get zero Tukey weight, the reduced system is nearly
    singular from iteration 7 on, and a PCG needs more than 3000 iterations per solve there
    (profiles/r04b_tukey_rough_start_pcg_iterations.txt) -- an iterative reduced solve is the wrong tool for that system, a direct one
    (the reference's, the oracle's) is not bothered.  Round 5: the HIP path hands such solves to its own exact solver
    (csrc/ba_direct.hip) and is pinned to the reference like the oracle, with the reference-vs-itself spread as the yardstick.
    Round 6: that solver is a sparse tile Cholesky (ordering + symbolic analysis once per structure, like the reference's
    SparseLinearSolver) without a size limit, and the same start is pinned at S2M size (s2m_lm10m_tukey: 4999 free poses)."""
    from cuba_amd.capi import HipSolver
    from oracle.oracle import OracleSolver
    make, rk, iters = rejected_trial_cases()[name]
    g = make()
    ref = ref_lm.run(g, rk, iters)
    fp = flatten(g)
    o = OracleSolver(fp, rk); ro = o.optimize(iters)
    assert ro["trials"].max() > 1, ro["trials"]                       # the case really rejects trials
    dev = {"oracle chi2": float(np.abs(ro["chi2"] / ref["chi2"] - 1).max()) if len(ro["chi2"]) == len(ref["chi2"]) else np.inf}
    est, trials = {}, {}
    for nm, a, b in zip("qtX", in_graph_order(fp, g, o.state()), (ref["q"], ref["t"], ref["Xw"])):
        est[f"oracle {nm}"] = float(np.abs(a - b).max())
    if name.endswith("_tukey"):
        # The reference accumulates with atomics: on this non-convex run last-bit differences of its own sums
        # are amplified along the rejected / re-tried steps, and landmarks whose observations all have zero Tukey weight are held by
        # the damping term alone.  So the yardstick is the reference against ITSELF: a second run of it, and the oracle must agree
        # with the first as well as that does (x 10), with 1e-9 / 1e-7 as floors.
        # (round 5: TWO further runs and the largest pairwise difference -- one extra run is a noisy sample of the spread, 1.2e-10 ... 2.6e-9 on
        # chi2 over four boxes, and the oracle's own 3.9e-9 then failed a 1.2e-9 bar -- and a floor of 1e-8 on chi2)
        refs = [ref, ref_lm.run(g, rk, iters), ref_lm.run(g, rk, iters)]
        pairs = [(refs[a], refs[b]) for a in range(3) for b in range(a + 1, 3)]
        self_chi = max(float(np.abs(a["chi2"] / b["chi2"] - 1).max()) if len(a["chi2"]) == len(b["chi2"]) else np.inf for a, b in pairs)
        self_est = {nm: max(float(np.abs(a[key] - b[key]).max()) for a, b in pairs) for nm, key in zip("qtX", ("q", "t", "Xw"))}
        # HIP path, default options: the solves whose PCG cannot finish within its budget are finished EXACTLY on the device
        # (csrc/ba_direct.hip, counter "exact_solve_fallbacks"), so the trajectory is the reference's -- same trials, same chi2
        # at the yardstick above -- and not a sequence of rejected "failed" solves (rounds 1-4)
        hip = {}
        for label, opts in (("hip tight", dict(pcg_tol=1e-11)), ("hip default", dict())):
            h = HipSolver(fp, rk, **opts); rh = h.optimize(iters)["chi2"]
            hip[label] = dict(chi=float(np.abs(rh / ref["chi2"] - 1).max()) if len(rh) == len(ref["chi2"]) else np.inf,
                              trials=h.counters()["lm_trials"], direct=h.counter("exact_solve_fallbacks"), failed=h.counter("exact_solve_failures"),
                              est={nm: float(np.abs(a - b).max()) for nm, a, b in zip("qtX", in_graph_order(fp, g, h.state()), (ref["q"], ref["t"], ref["Xw"]))},
                              median_t=float(np.median(np.abs(in_graph_order(fp, g, h.state())[1] - ref["t"]).max(1))))
            h.close()
        print(f"\n[{name}] trials per iteration {ro['trials'].tolist()}: oracle vs the reference's own optimiser "
              + ", ".join(f"{k} {v:.2e}" for k, v in {**dev, **est}.items())
              + f"; the reference vs two further runs of itself (largest pairwise difference): chi2 {self_chi:.2e}, " + ", ".join(f"{k} {v:.2e}" for k, v in self_est.items())
              + "; " + "; ".join(f"{k}: chi2 {v['chi']:.2e}, {v['trials']} trials, {v['direct']} exact solves ({v['failed']} failed), "
                                 + ", ".join(f"{a} {b:.2e}" for a, b in v["est"].items()) for k, v in hip.items()))
        assert dev["oracle chi2"] <= max(1e-8, 10 * self_chi), (dev, self_chi)
        # estimates: vertices whose observations all carry zero Tukey weight sit where the damping term alone leaves them and follow every
        # last-bit difference of the sums (the oracle is 8e-3 m from the reference on a box where two reference runs are 3e-4 m apart, and
        # 1e-3 m where they are 5e-3 m apart): floors of 2e-2 m / 1e-4 on the quaternions, five times that at the default tolerance
        floor = {"q": 1e-4, "t": 2e-2, "X": 2e-2}
        chi_floor = {"hip tight": 1e-8, "hip default": 1e-6}
        if name == "s2m_lm10m_tukey":
            # S2M size: the run agrees with the oracle to 3e-11 on chi2 for ten iterations (most vertices bit for bit: median difference of the
            # final estimates 0), then the last four iterations -- three of them with a rejected trial -- multiply every difference by ~10 each:
            # the reference against itself ends 1e-9 ... 9e-9 apart (0.12 m on the poses of one weakly held stretch of the trajectory, ~600 of
            # 5000 poses), the oracle 5e-10 / 0.06 m from it, the HIP path -- same mathematics, other summation orders and formulas, every one
            # of these solves exact -- 4e-8 ... 7e-8 / 0.7 m (default tolerance 1e-7 / 2.5 m); profiles/r06d_tukey_s2m.txt.  The bars are the
            # north star's 1e-6 on chi2 at the default tolerance and 3e-7 at the tight one, and the floppy stretch's positions within metres.
            floor = {"q": 2e-3, "t": 1.5, "X": 5e-2}
            chi_floor = {"hip tight": 3e-7, "hip default": 1e-6}
        for nm in "qtX":
            assert est[f"oracle {nm}"] <= max(floor[nm], 10 * self_est[nm]), (nm, est, self_est)
        for label, v in hip.items():
            assert v["chi"] <= max(chi_floor[label], 10 * self_chi), (label, v, self_chi)
            assert v["median_t"] <= 1e-5, (label, v)          # (the bulk of the trajectory is the reference's to far better than the bars above)
            assert v["trials"] == int(ro["trials"].sum()), (label, v["trials"], ro["trials"])
            assert v["direct"] >= 1 and v["failed"] == 0, (label, v)
            for nm in "qtX":
                assert v["est"][nm] <= max((1 if "tight" in label else 5) * floor[nm], 10 * self_est[nm]), (label, nm, v["est"], self_est)
        return
    for label, opts in (("hip tight", dict(pcg_tol=1e-11)), ("hip default", dict())):
        h = HipSolver(fp, rk, **opts); rh = h.optimize(iters)["chi2"]
        dev[label + " chi2"] = float(np.abs(rh / ref["chi2"] - 1).max()) if len(rh) == len(ref["chi2"]) else np.inf
        trials[label] = h.counters()["lm_trials"]
        for nm, a, b in zip("qtX", in_graph_order(fp, g, h.state()), (ref["q"], ref["t"], ref["Xw"])):
            est[f"{label} {nm}"] = float(np.abs(a - b).max())
        assert h.pcg_history()[1] == 0
        h.close()
    print(f"\n[{name}] trials per iteration {ro['trials'].tolist()} vs the reference's own optimiser: "
          + ", ".join(f"{k} {v:.2e}" for k, v in {**dev, **est}.items()))
    assert dev["oracle chi2"] <= 1e-9 and dev["hip tight chi2"] <= 1e-8 and dev["hip default chi2"] <= 1e-6, dev
    assert trials["hip tight"] == trials["hip default"] == int(ro["trials"].sum()), (trials, ro["trials"])
    for k, v in est.items():
        lim = 1e-7 if k.startswith("oracle") else 1e-6 if "tight" in k else 1e-5
        assert v <= lim, (k, v, lim)
