"""round 5: wall of optimize(10) on the large shapes (bench protocol: warm-up iteration, perturbed starts) with and without an option.
   python scripts/r05/shapes_time.py s2m spmv_upper=0     (one handle per process)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from cuba_amd.capi import HipSolver
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_named
RK = ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815))))
shape = sys.argv[1]
opts = {k: float(v) for k, v in (a.split("=") for a in sys.argv[2:] if "=" in a)}
g = synth_named(shape)
if "sortlm" in sys.argv:
    # landmark ids re-assigned in the order of the first (then the last) pose that observes them: what creation-ordered map points look like
    lut = np.zeros(int(g.lm_ids.max()) + 1, dtype=np.int64); lut[g.lm_ids] = np.arange(g.nlandmarks)
    vl = lut[np.concatenate([g.mono_vl, g.stereo_vl])]; vp = np.concatenate([g.mono_vp, g.stereo_vp])
    first = np.full(g.nlandmarks, 1 << 40); np.minimum.at(first, vl, vp)
    last = np.zeros(g.nlandmarks, dtype=np.int64); np.maximum.at(last, vl, vp)
    rank = np.empty(g.nlandmarks, dtype=np.int64); rank[np.lexsort((last, first))] = np.arange(g.nlandmarks)
    new_ids = int(g.lm_ids.min()) + rank
    g.mono_vl = new_ids[lut[g.mono_vl]]; g.stereo_vl = new_ids[lut[g.stereo_vl]]; g.lm_ids = new_ids
fp = flatten(g)
h = HipSolver(fp, RK, **opts)
chi = h.optimize(10)["chi2"]
h.set_state(fp.q, fp.t, fp.Xw); h.optimize(1)
n = bench.prepare_slots(h, fp, h.state(), 4, seed=1000)
def timed(slot):
    h.restore_state(slot); c0 = h.counters(); t = time.perf_counter(); h.optimize(10); dt = time.perf_counter() - t
    return 1e3 * dt, h.counters()["pcg_iterations"] - c0["pcg_iterations"]
timed(n)
w = [timed(1 + k) for k in range(3)]
timed(0); r = [timed(0) for _ in range(3)]
kt = h.time_kernels(10)
print(f"{shape} {opts}: perturbed {np.median([x[0] for x in w]):.2f} ms ({int(np.median([x[1] for x in w]))} PCG iterations), replay {np.median([x[0] for x in r]):.2f} ms; "
      f"chi2[-1] {chi[-1]:.9e}; kernels (us): " + ", ".join(f"{k} {1e3 * v:.1f}" for k, v in kt.items() if v > 0), flush=True)
h.close()
