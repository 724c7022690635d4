"""Per-kernel device times only (no converged solve needed): for timing-only experiment builds whose arithmetic is wrong on purpose.
   CUBA_HIP_LIB_F64=... python scripts/r04/spmv_timing_only.py s2m"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from cuba_amd.capi import HipSolver
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
shape = sys.argv[1]
fp = flatten(synth_named(shape))
h = HipSolver(fp, RK, pcg_max_iter=8, pcg_accept_unconverged=1); h.build_structure()
try:
    h.optimize(1)
except Exception as e:
    print("optimize:", e)
kt = h.time_kernels(20)
print(shape, os.path.basename(os.environ.get("CUBA_HIP_LIB_F64", "default")), "  ".join("%s %.2f us" % (k, v * 1e3) for k, v in kt.items() if v > 0), flush=True)
