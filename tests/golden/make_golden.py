"""Generates tests/golden/lm_trajectories.json with the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`).  These are oracle outputs, NOT reference outputs: the reference
(CUDA + cuSOLVER + Eigen) cannot run in this container and its README chi2 table needs the absent KITTI
dataset.  They pin the oracle and the HIP path against regressions on fixed seeded inputs."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cuba_amd.graph import flatten  # noqa: E402
from cuba_amd.synth import synth_ba  # noqa: E402
from oracle.oracle import OracleSolver  # noqa: E402

H = [[1, float(np.sqrt(5.991))], [1, float(np.sqrt(7.815))]]
CASES = [
    dict(name="tiny_none", graph=dict(P=10, L=60, E=200, seed=11), robust=[[0, 0.0], [0, 0.0]], iterations=6),
    dict(name="small_huber", graph=dict(P=40, L=600, E=2400, seed=1), robust=H, iterations=10),
    dict(name="small_tukey", graph=dict(P=40, L=600, E=2400, seed=1), robust=[[2, 4.0], [2, 5.0]], iterations=6),
    dict(name="mid_huber", graph=dict(P=120, L=6000, E=24000, seed=9), robust=H, iterations=10),
    dict(name="mono_only", graph=dict(P=30, L=400, E=1600, seed=4, stereo_frac=0.0), robust=H, iterations=5),
]
out = {"generator": "oracle/ba_oracle.cpp via tests/golden/make_golden.py", "cases": []}
for c in CASES:
    fp = flatten(synth_ba(**c["graph"]))
    res = OracleSolver(fp, tuple(map(tuple, c["robust"]))).optimize(c["iterations"])
    q, t, X = OracleSolver(fp, tuple(map(tuple, c["robust"]))).state()
    c = dict(c); c["chi2"] = [float(v) for v in res["chi2"]]; c["lambdas"] = [float(v) for v in res["lambdas"]]
    out["cases"].append(c)
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lm_trajectories.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote", len(out["cases"]), "cases")
