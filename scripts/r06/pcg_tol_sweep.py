"""chi2 trajectory against the committed oracle golden, PCG iterations and wall of a 10-iteration run as a function of pcg_tol"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cuba_amd import capi
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_named
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "baseline_shapes_chi2.json")))["shapes"]
rk = ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815))))
for shape in sys.argv[1:] or ["kitti07", "kitti00", "s2m"]:
    fp = flatten(synth_named(shape))
    g = np.array(gold[shape]["chi2"])
    for tol in (1e-7, 3e-7, 1e-6, 3e-6, 1e-5, 1e-4):
        h = capi.HipSolver(fp, rk, pcg_tol=tol)
        h.snapshot_state(1)
        r = h.optimize(10)                      # (first run on the structure)
        ms = []
        for _ in range(5):
            h.restore_state(1)
            t0 = time.perf_counter(); r = h.optimize(10); ms.append((time.perf_counter() - t0) * 1e3)
        c = np.array(r["chi2"]); n = min(len(c), len(g))
        it = h.counters()["pcg_iterations"] // 6
        print(f"{shape:8s} pcg_tol {tol:.0e}: chi2 max rel diff vs golden {np.max(np.abs(c[:n] - g[:n]) / g[:n]):.2e} (final {abs(c[n-1]-g[n-1])/g[n-1]:.2e})  pcg iterations {it}  wall min {min(ms):.3f} ms", flush=True)
        h.close()
