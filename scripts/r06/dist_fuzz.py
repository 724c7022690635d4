"""Randomised sweep of the landmark-partitioned mode (not part of the suite): the native driver (libcuba_hip_dist.so) with 2-8 ranks emulated as host threads
on one GPU (in-process communicator), random graphs, robust kernels, fixed vertices, whole and ranged uploads, against the single-handle run."""
import os, sys, copy, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from cuba_amd.capi import HipSolver
from cuba_amd.dist import NativeDist, ThreadComm, landmark_ranges
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_ba
from test_dist import _values_only_inside

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(5)
KINDS = [((0, 0.0), (0, 0.0)), ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815)))), ((2, 4.0), (2, 5.0))]
bad = 0; done = 0
for rd in range(rounds):
    P = int(rng.integers(20, 400)); L = int(rng.integers(8 * P, 40 * P)); E = int(L * rng.uniform(2.6, 5.0))
    try:
        g = copy.deepcopy(synth_ba(P, L, E, seed=int(rng.integers(1 << 30)), stereo_frac=float(rng.choice([0.3, 0.85, 1.0])), loop_closure=bool(rng.integers(2))))
    except (RuntimeError, ValueError):
        continue
    if rng.random() < 0.4: g.pose_fixed[rng.choice(P, size=int(rng.integers(1, 4)), replace=False)] = True
    if rng.random() < 0.3: g.lm_fixed[rng.choice(g.nlandmarks, size=g.nlandmarks // 10, replace=False)] = True
    fp = flatten(g)
    key = fp.eP.astype(np.int64) * (fp.Lt + 1) + fp.eL
    if len(np.unique(key)) != fp.E or fp.Pt == fp.Pf or fp.Pf == 0 or fp.Lf == 0: continue
    rk = KINDS[int(rng.integers(len(KINDS)))]; world = int(rng.choice([2, 3, 4, 8])); iters = int(rng.integers(3, 9)); ranged = bool(rng.integers(2))
    single = HipSolver(fp, rk); want = single.optimize(iters)["chi2"]; q1, t1, X1 = single.state()
    comms = ThreadComm.create(world); out, err = [None] * world, []
    def work(c):
        try:
            if ranged:
                h = HipSolver(None, rk); r = landmark_ranges(fp.eL, fp.Lt, world)[c.rank]
                h.set_graph(_values_only_inside(fp, r), landmark_range=r)
            else:
                h = HipSolver(fp, rk)
            d = NativeDist(h, fp, c.rank, world, comm=c)
            chi2 = d.optimize(iters); out[c.rank] = (chi2, d.complete_solution()); d.close()
        except Exception as e:
            err.append(e); c.s.barrier.abort()
    th = [threading.Thread(target=work, args=(c,)) for c in comms]
    [t.start() for t in th]; [t.join() for t in th]
    done += 1
    label = f"round {rd}: P {fp.Pt} (free {fp.Pf}) L {fp.Lt} (free {fp.Lf}) E {fp.E} kernels {rk[0][0]} ranks {world} iters {iters} ranged {ranged}"
    if err:
        bad += 1; print("FAIL (exception)", label, err[0], flush=True); continue
    same_replicas = all(np.array_equal(out[0][0], o[0]) for o in out[1:])
    chi = max(float(np.max(np.abs(o[0] - want) / want)) if len(o[0]) == len(want) else 1.0 for o in out)
    est = [max(float(np.abs(a - b).max()) for o in out for a, b in [(o[1][k], (q1, t1, X1)[k])]) for k in range(3)]
    if not same_replicas or chi > 1e-6 or est[0] > 1e-6 or est[1] > 1e-4 or est[2] > 1e-4:
        bad += 1; print("FAIL", label, f"replicas identical {same_replicas} chi2 {chi:.2e} worst q {est[0]:.2e} t {est[1]:.2e} X {est[2]:.2e}", flush=True)
print(f"partitioned mode: {done} random graphs x 2-8 emulated ranks against the single-handle run: {bad} failures", flush=True)
