"""BASELINE.json configs 3 and 5 (single GPU): S2M (5k/500k/2M) and G4M (10k/1M/4M) shapes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from cuba_amd.capi import HipSolver
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
for name in [a for a in sys.argv[1:] if not a.startswith("--")] or ["s2m"]:
    t = time.time(); fp = flatten(synth_named(name)); tg = time.time() - t
    t = time.time(); h = HipSolver(fp, RK); h.build_structure(); ts = time.time() - t
    q0, t0, X0 = h.state()
    h.optimize(10); h.set_state(q0, t0, X0); c0 = h.counters()
    t = time.perf_counter(); chi2 = h.optimize(10)["chi2"]; dt = time.perf_counter() - t
    c = h.counters()
    print(name, "P/L/E", fp.Pt, fp.Lt, fp.E, "gen %.1fs setup %.2fs" % (tg, ts), "10 iters %.1f ms -> %.1f M edge-it/s" % (dt * 1e3, fp.E * len(chi2) / dt / 1e6),
          "pcg", c["pcg_iterations"] - c0["pcg_iterations"], "nblk", c["hsc_blocks"], "nmul", c["schur_products"])
    print("  chi2", chi2[0], "->", chi2[-1], "monotone", bool(np.all(np.diff(chi2) < 0)))
    print("  kernels", {k: round(v * 1e3, 1) for k, v in h.time_kernels(5).items()})
    if "--oracle" in sys.argv:
        from oracle.oracle import OracleSolver
        t = time.time(); ref = OracleSolver(fp, RK).optimize(10)["chi2"]; print("  oracle %.1fs chi2 max rel diff %.2e" % (time.time() - t, np.max(np.abs(chi2 - ref) / ref)))
    h.close()
