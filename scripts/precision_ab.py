"""fp64 / mixed precision / all-fp32 side by side (the reference's USE_FLOAT32 option src/scalar.h:25-29):
wall of 10-iteration runs, per-iteration chi2 and final estimates against the exact-solve CPU oracle.
   python scripts/precision_ab.py kitti00"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from cuba_amd.capi import HipSolver
from oracle.oracle import OracleSolver
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
shape = sys.argv[1] if len(sys.argv) > 1 else "kitti00"
fp = flatten(synth_named(shape))
o = OracleSolver(fp, RK); ref = o.optimize(10)["chi2"]; rs = o.state()
rm = lambda a, b: float(np.sqrt(((a - b) ** 2).sum(1).mean()))
print(f"{shape}: P {fp.Pt} L {fp.Lt} E {fp.E}")
for name, prec, opts in (("fp64", "f64", {}), ("mixed (fp32 records + per-edge Schur arithmetic, fp64 sums / reduced system / PCG)", "f64", dict(mixed_precision=1)),
                         ("all fp32 (USE_FLOAT32 build)", "f32", {})):
    h = HipSolver(fp, RK, precision=prec, **opts); h.build_structure()
    q0, t0, X0 = h.state()
    got = h.optimize(10)["chi2"]; st = h.state()
    m = min(len(got), len(ref))
    ts = []
    for _ in range(8):
        h.set_state(q0, t0, X0)
        t = time.perf_counter(); h.optimize(10); ts.append(time.perf_counter() - t)
    kt = h.time_kernels(20)
    print(f"{name}\n   10-iter wall min {min(ts)*1e3:.3f} ms   linearize+Schur {kt['linearize_schur']*1e3:.1f} us   iterations done {len(got)}   "
          f"PCG its {int(np.abs(h.pcg_history()[0][-10:]).sum())}\n   chi2 max rel diff vs oracle {np.abs(got[:m]/ref[:m]-1).max():.2e}   "
          f"RMSE q {rm(st[0], rs[0]):.2e} t {rm(st[1], rs[1]):.2e} X {rm(st[2], rs[2]):.2e}", flush=True)
