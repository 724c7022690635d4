"""Generates tests/golden/config4_seeds_chi2.json: the CPU oracle's chi2 trajectory (10 LM iterations from the generator's initial
guess, Huber kernels of the samples) on the eight graphs of BASELINE.json configs[3] -- KITTI-00-sized, seeds 100..107, the graphs
`bench.py --gpus N` gives to rank 0..7 -- plus the RMS of the oracle's final estimates as a cheap fingerprint.  Oracle outputs, NOT
reference outputs (the oracle itself is pinned to the reference's own optimiser at this shape by tests/test_ref_lm.py).  Every rank
of the weak-scaling bench compares its own 10-iteration run with its entry (`chi2_max_rel_diff_vs_golden` per rank in the line);
tests/test_gpu_configs.py::test_config4_graphs_follow_the_oracle re-derives every entry with the live oracle on the GPU box.
Run from the repo root: `python tests/golden/make_golden_config4.py`."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cuba_amd.graph import flatten  # noqa: E402
from cuba_amd.synth import synth_named  # noqa: E402
from oracle.oracle import OracleSolver  # noqa: E402

RK = ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815))))
PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config4_seeds_chi2.json")
SEEDS = list(range(100, 108))

out = {"generator": "oracle/ba_oracle.cpp (1 thread) via tests/golden/make_golden_config4.py", "shape": "kitti00",
       "robust": [list(RK[0]), list(RK[1])], "iterations": 10, "seeds": {}}
for seed in SEEDS:
    fp = flatten(synth_named("kitti00", seed=seed))
    t = time.time()
    o = OracleSolver(fp, RK)
    r = o.optimize(10)
    q, tt, X = o.state()
    out["seeds"][str(seed)] = {"P": fp.Pt, "L": fp.Lt, "E": fp.E, "chi2": [float(v) for v in r["chi2"]],
                               "trials": [int(v) for v in r["trials"]], "oracle_seconds": round(time.time() - t, 2),
                               "final_rms": {"q": float(np.sqrt((q * q).mean())), "t": float(np.sqrt((tt * tt).mean())),
                                             "X": float(np.sqrt((X * X).mean()))}}
    print(seed, out["seeds"][str(seed)]["oracle_seconds"], "s", r["chi2"][0], "->", r["chi2"][-1], flush=True)
with open(PATH, "w") as f:
    json.dump(out, f, indent=1)
