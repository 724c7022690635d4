"""Soak test of the asynchronous pieces (helper-thread hipGraph instantiation / destruction, overlapped coarse inversion, structure cache):
one handle fed alternating topologies and estimates, a second handle created and destroyed around it, results compared with a fresh handle's."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from cuba_amd.synth import synth_ba
from cuba_amd.graph import flatten
from cuba_amd.capi import HipSolver
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
rng = np.random.default_rng(7)
fps = [flatten(synth_ba(P=p, L=l, E=4 * l, seed=s)) for p, l, s in ((60, 3000, 1), (90, 5000, 2), (40, 1500, 3), (300, 30000, 4))]
ref = []
for fp in fps:
    h = HipSolver(fp, RK); r = h.optimize(6)["chi2"]; h.close(); ref.append(np.array(r))
h = HipSolver(fps[0], RK)
t0 = time.time(); n = 0; worst = 0.0
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
while time.time() - t0 < budget:
    i = int(rng.integers(len(fps)))
    fp = fps[i]
    h.set_graph(fp)
    its = int(rng.integers(1, 7))
    got = np.array(h.optimize(its)["chi2"])
    d = float(np.max(np.abs(got - ref[i][:len(got)]) / ref[i][:len(got)]))
    worst = max(worst, d)
    assert d < 1e-7, (n, i, its, d)
    if n % 7 == 0:
        h2 = HipSolver(fps[(i + 1) % len(fps)], RK); h2.optimize(2); h2.close()
    if n % 11 == 0:
        h.snapshot_state(); a = np.array(h.optimize(2)["chi2"]); h.restore_state(); b = np.array(h.optimize(2)["chi2"])
        assert np.allclose(a, b, rtol=1e-9), (n, a, b)
    n += 1
h.close()
print("stress ok: %d optimise calls over %d topologies, worst chi2 rel diff vs a fresh handle %.2e" % (n, len(fps), worst))
