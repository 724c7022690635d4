"""Per-kernel device times (cuba_hip_time_kernels) + wall of 10-iteration runs, one line: for A/B runs of builds or options.
   [CUBA_HIP_LIB_F64=path/to/experiment.so] python scripts/kernel_times.py kitti00 [option=value ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from cuba_amd.capi import HipSolver
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
shape = sys.argv[1] if len(sys.argv) > 1 else "kitti00"
opts = dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in sys.argv[2:])
fp = flatten(synth_named(shape))
h = HipSolver(fp, RK, **opts); h.build_structure()
h.optimize(1)
q0, t0, X0 = h.state()
h.optimize(10)
ts = []
for rep in range(12):
    h.set_state(q0, t0, X0)
    t = time.perf_counter(); chi2 = h.optimize(10)["chi2"]; ts.append(time.perf_counter() - t)
it, bad = h.pcg_history()
kt = h.time_kernels(20)
print("%s %s lib=%s  10-iter min %.3f ms median %.3f ms  pcg its/run %d  chi2 %.6f | " % (
    shape, opts, os.path.basename(os.environ.get("CUBA_HIP_LIB_F64", "default")), min(ts) * 1e3, np.median(ts) * 1e3, int(np.abs(it[-10:]).sum()), chi2[-1])
    + "  ".join("%s %.2f us" % (k, v * 1e3) for k, v in kt.items() if v > 0), flush=True)
