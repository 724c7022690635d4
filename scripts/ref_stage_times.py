"""Same-node stage baseline from the REFERENCE ITSELF (round-2 verdict, item 8).

The reference's own device functions (namespace cuba::gpu of /root/reference/src/cuda_block_solver.cu, compiled in place into
oracle/_ref/libcuba_ref_kernels.so) are timed stage by stage on this MI355X at a BASELINE shape, HIP events around each call
exactly as CudaBlockSolver makes it (blocking scalar read-backs included), next to this library's kernels for the same stages.
It is the only like-for-like number rows a1-a5 / a8 of SURVEY section 8 can have: same hardware, same graph, same arithmetic.
The reference's reduced solve (cuSOLVER) cannot run here and is absent from the table.

usage: python scripts/ref_stage_times.py [shape] [reps]      (run on the GPU box; writes a table to stdout)
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cuba_amd.capi import HipSolver          # noqa: E402
from cuba_amd.graph import flatten           # noqa: E402
from cuba_amd.synth import synth_named       # noqa: E402
from oracle import ref_kernels               # noqa: E402

RK = ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815))))


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "kitti00"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    fp = flatten(synth_named(shape))
    h = HipSolver(fp, RK, pose_reorder=0)
    md = h.max_diagonal()
    lam = 1e-5 * md
    rp, ci = h.hsc_structure()
    lib = C.CDLL(ref_kernels.LIB)
    lib.ref_set_timing(C.c_int(reps))
    ref_kernels.run_trial(fp, RK, rp, ci, lam, np.zeros(6 * fp.Pf))
    ms = (C.c_double * 8)()
    lib.ref_get_stage_ms(ms)
    lib.ref_set_timing(C.c_int(0))
    ref = dict(zip(("computeActiveErrors x2", "fillZero + constructQuadraticForm x2", "maxDiagonal x2", "addLambda + restoreDiagonal x2",
                    "computeBschure", "computeHschure", "schurComplementPost", "computeScale + updatePoses + updateLandmarks"), list(ms)))
    h.set_lambda(lam)
    ours = h.time_kernels(reps=max(reps, 20))
    print(f"shape {shape}: P/L/E {fp.Pt}/{fp.Lt}/{fp.E}, Hsc blocks {len(ci)}; average of {reps} calls, milliseconds on the MI355X")
    print("reference (its own kernels, compiled in place)            | this library")
    rows = [
        ("computeActiveErrors x2", "residual_chi2", "a1"),
        ("fillZero + constructQuadraticForm x2", None, "a2"),
        ("computeBschure", None, "a4"),
        ("computeHschure", None, "a5"),
        ("schurComplementPost", "back_substitute", "a8"),
    ]
    for rname, oname, row in rows:
        o = f"{oname} {ours[oname]:.4f}" if oname else ""
        print(f"  [{row}] {rname:42s} {ref[rname]:9.4f} | {o}")
    lin = ref["fillZero + constructQuadraticForm x2"] + ref["addLambda + restoreDiagonal x2"] / 2 + ref["computeBschure"] + ref["computeHschure"]
    print(f"  [a2+a3+a4+a5] build system + damping + Schur complement   {lin:9.4f} | linearize_schur {ours['linearize_schur']:.4f}  ({lin / ours['linearize_schur']:.1f}x)")
    for rname in ("maxDiagonal x2", "addLambda + restoreDiagonal x2", "computeScale + updatePoses + updateLandmarks"):
        print(f"  [--] {rname:42s} {ref[rname]:9.4f} |")
    print(f"  ratios: errors {ref['computeActiveErrors x2'] / ours['residual_chi2']:.1f}x, "
          f"back-substitution {ref['schurComplementPost'] / ours['back_substitute']:.1f}x")


if __name__ == "__main__":
    main()
