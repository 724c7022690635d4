"""Why does a shape's 10-iteration wall differ between scripts/big_shapes.py and bench.py's `shapes` leg?  Times optimize(10) on one shape
for {the handle's own stream, torch's current stream} x {from the generator's initial guess, from the state after a 1-iteration warm-up}.
   python scripts/shape_protocol_ab.py [shape]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cuba_amd.capi import HipSolver
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_named

RK = ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815))))
shape = sys.argv[1] if len(sys.argv) > 1 else "kitti07"
fp = flatten(synth_named(shape))
for stream_name, stream in (("own stream", None), ("torch current stream", torch.cuda.current_stream().cuda_stream)):
    for start in ("initial", "warm"):
        h = HipSolver(fp, RK, stream=stream)
        h.optimize(10)
        h.set_state(fp.q, fp.t, fp.Xw)
        if start == "warm":
            h.optimize(1)
        q1, t1, X1 = h.state()
        walls, its = [], []
        for _ in range(5):
            h.set_state(q1, t1, X1); c0 = h.counters()
            torch.cuda.synchronize(); t = time.perf_counter(); r = h.optimize(10)["chi2"]; torch.cuda.synchronize()
            walls.append((time.perf_counter() - t) * 1e3); c1 = h.counters()
            its.append((c1["pcg_iterations"] - c0["pcg_iterations"], c1["pcg_iterations_enqueued"] - c0["pcg_iterations_enqueued"], c1["lm_trials"] - c0["lm_trials"], c1["coarse_refreshes"] - c0["coarse_refreshes"]))
        print(f"{shape} {stream_name:22s} start {start:8s} walls ms {[round(w, 2) for w in walls]}  (pcg, enqueued, trials, refreshes) {its[-1]}  iterations per solve {h.pcg_history()[0][-10:].tolist()}", flush=True)
        h.close()
