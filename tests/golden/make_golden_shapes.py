"""Generates tests/golden/baseline_shapes_chi2.json: the CPU oracle's chi2 trajectory (10 LM iterations from the generator's
initial guess, Huber kernels of the samples) on every BASELINE.json shape -- kitti07, kitti00, s2m, g4m.  Oracle outputs, NOT
reference outputs (see make_golden.py); bench.py checks its driver-timed runs of the large shapes against them without paying
the oracle's CPU time inside the bench (the G4M run takes minutes of one core), and tests/test_oracle_system.py re-derives the
KITTI-07 entry on every CPU run.  Run from the repo root: `python tests/golden/make_golden_shapes.py [shape ...]`."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cuba_amd.graph import flatten  # noqa: E402
from cuba_amd.synth import SHAPES, synth_named  # noqa: E402
from oracle.oracle import OracleSolver  # noqa: E402

RK = ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815))))
PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "baseline_shapes_chi2.json")

out = json.load(open(PATH)) if os.path.exists(PATH) else {"generator": "oracle/ba_oracle.cpp (1 thread) via tests/golden/make_golden_shapes.py",
                                                          "robust": [list(RK[0]), list(RK[1])], "iterations": 10, "shapes": {}}
for name in sys.argv[1:] or list(SHAPES):
    fp = flatten(synth_named(name))
    t = time.time()
    r = OracleSolver(fp, RK).optimize(10)
    out["shapes"][name] = {"P": fp.Pt, "L": fp.Lt, "E": fp.E, "seed": SHAPES[name]["seed"], "chi2": [float(v) for v in r["chi2"]],
                           "trials": [int(v) for v in r["trials"]], "oracle_seconds": round(time.time() - t, 2)}
    print(name, out["shapes"][name]["oracle_seconds"], "s", r["chi2"][0], "->", r["chi2"][-1], flush=True)
    with open(PATH, "w") as f:
        json.dump(out, f, indent=1)
