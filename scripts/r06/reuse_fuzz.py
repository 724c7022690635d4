"""Randomised sweep over the life of a handle (not part of the suite): one handle is given a random sequence of graphs of changing size (plain and
two-step uploads, re-uploads of the same graph with and without the unchanged-graph hint, changed measurements, snapshots), optimises after each step, and
every run is compared bit for bit with a fresh handle's run of the same graph from the same estimates (heuristics = 0 on both sides: the run-to-run
memories of a handle are what a fresh one cannot have), and with the oracle at the stated bars.  Then batches: random subsets of same-class graphs with
different kernels and iteration counts against their solo runs, bit for bit; and the fp32 library against the fp64 one."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cuba_amd import capi
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_ba
from oracle.oracle import OracleSolver

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(77)
KINDS = [((0, 0.0), (0, 0.0)), ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815)))), ((2, 4.0), (2, 5.0))]
def draw_graph(lo=5, hi=450):
    while True:
        P = int(rng.integers(lo, hi)); L = int(rng.integers(6 * P, 30 * P + 40)); E = int(L * rng.uniform(2.6, 5.0))
        try:
            g = synth_ba(P, L, E, seed=int(rng.integers(1 << 30)), stereo_frac=float(rng.choice([0.3, 0.85, 1.0])), loop_closure=bool(rng.integers(2)))
        except (RuntimeError, ValueError):
            continue
        fp = flatten(g)
        key = fp.eP.astype(np.int64) * (fp.Lt + 1) + fp.eL
        if len(np.unique(key)) == fp.E and fp.Pt > fp.Pf and fp.Pf > 0 and fp.Lf > 0: return fp
bad = 0; checks = 0
def same(a, b): return len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))
for rd in range(rounds):
    rk = KINDS[int(rng.integers(len(KINDS)))]
    opts = dict(heuristics=0)
    if rng.random() < 0.3: opts["device_setup"] = 0
    h = capi.HipSolver(None, rk, **opts)
    fp = None
    for step in range(int(rng.integers(4, 9))):
        act = rng.choice(["new", "new", "same", "same_hint", "values", "state"]) if fp is not None else "new"
        if act == "new": fp = draw_graph()
        elif act == "values":
            fp = copy.copy(fp); fp.meas = fp.meas + rng.normal(0, 0.2, fp.meas.shape)
        if act == "same_hint": h.hint_unchanged(True, True)
        if act == "state" and fp is not None:
            # estimates moved by the caller between runs (set_state), no upload
            q, t, X = h.state(); X = X + rng.normal(0, 0.01, X.shape); h.set_state(q, t, X)
            start = (q, t, X)
        else:
            h.set_graph(fp, two_step=bool(rng.integers(2)))
            start = None
        it = int(rng.integers(1, 6))
        r = h.optimize(it)["chi2"]; sh = h.state()
        f = capi.HipSolver(fp, rk, **opts)
        if start is not None: f.set_state(*start)
        rf = f.optimize(it)["chi2"]; sf = f.state()
        checks += 1
        if not (np.array_equal(r, rf) and same(sh, sf)):
            bad += 1
            print(f"FAIL round {rd} step {step} action {act} P {fp.Pt} L {fp.Lt} E {fp.E} kernels {rk[0][0]} iters {it} opts {opts}: reused handle vs fresh handle chi2 {r} vs {rf}; state equal {same(sh, sf)}", flush=True)
        if start is None:
            o = OracleSolver(fp, rk); ro = o.optimize(it)["chi2"]
            if len(ro) != len(r) or np.max(np.abs(r - ro) / ro) > 1e-6:
                bad += 1; print(f"FAIL round {rd} step {step} vs oracle: {r} {ro}", flush=True)
        f.close()
    h.close()
print(f"handle life: {checks} runs compared with fresh handles (bit for bit) and the oracle: {bad} failures", flush=True)
# ---- batches of same-class graphs
badb = 0; nb = 0
for rd in range(rounds):
    n = int(rng.integers(2, 7))
    lo = int(rng.choice([60, 250, 700])); fps = [draw_graph(lo, lo + 60) for _ in range(n)]
    rks = [KINDS[int(rng.integers(len(KINDS)))] for _ in range(n)]
    it = int(rng.integers(2, 7))
    solo = []
    for fp, rk in zip(fps, rks):
        s = capi.HipSolver(fp, rk); solo.append((s.optimize(it)["chi2"], s.state())); s.close()
    hs = [capi.HipSolver(fp, rk) for fp, rk in zip(fps, rks)]
    res, batched = capi.optimize_batch(hs, it)
    ok = all(np.array_equal(r, s[0]) and same(hh.state(), s[1]) for r, s, hh in zip(res, solo, hs))
    # a second batch on the same handles from where the first one ended, against solo continuation
    res2, _ = capi.optimize_batch(hs, 2)
    ok2 = True
    for fp, rk, s0, r2, hh in zip(fps, rks, solo, res2, hs):
        s = capi.HipSolver(fp, rk); s.optimize(it); c = s.optimize(2)["chi2"]; ok2 = ok2 and np.array_equal(c, r2) and same(s.state(), hh.state()); s.close()
    nb += 1
    if not (ok and ok2):
        badb += 1; print(f"FAIL batch round {rd}: {n} graphs around {lo} poses, iters {it}, first batch identical {ok}, continuation identical {ok2}, {batched} solves batched", flush=True)
    for hh in hs: hh.close()
print(f"batches: {nb} random batches (2-6 graphs, mixed kernels) against solo runs, first run and continuation: {badb} failures", flush=True)
# ---- fp32 library
bad32 = 0
for rd in range(rounds):
    fp = draw_graph(20, 300); rk = KINDS[1]
    a = capi.HipSolver(fp, rk).optimize(6)["chi2"]
    b = capi.HipSolver(fp, rk, precision="f32").optimize(6)["chi2"]
    n = min(len(a), len(b)); d = float(np.max(np.abs(a[:n] - b[:n]) / a[:n]))
    if n < 4 or d > 1e-3:
        bad32 += 1; print(f"FAIL fp32 round {rd}: P {fp.Pt} chi2 rel diff {d:.2e} lengths {len(a)} {len(b)}", flush=True)
print(f"fp32 library: {rounds} graphs, chi2 within 1e-3 of the fp64 run: {bad32} failures", flush=True)
