"""round 5: several live solver handles in one process (verdict item 4).  (1) what an IDLE second handle costs the first; (2) the second
handle's own speed; (3) 2 / 4 graphs optimised concurrently from distinct host threads on ONE GPU: per-graph wall, aggregate rate,
bit-identity with the solo run."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cuba_amd.capi import HipSolver
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_named
RK = ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815))))
shape = sys.argv[1] if len(sys.argv) > 1 else "kitti00"
OPTS = {k: float(v) for k, v in (a.split("=") for a in sys.argv[2:] if "=" in a)}
fp = flatten(synth_named(shape))
E = fp.E

def make():
    h = HipSolver(fp, RK, **OPTS); h.optimize(1); h.snapshot_state(); h.optimize(10); return h

if "loop" in sys.argv:
    # one handle, optimise for a fixed wall-clock window (two PROCESSES of this mode side by side: is the interference in-process?)
    h = make(); ts = []; t_end = time.time() + 6.0
    while time.time() < t_end:
        h.restore_state(); t = time.perf_counter(); h.optimize(10); ts.append(time.perf_counter() - t)
    print(f"{shape} loop pid {os.getpid()}: {len(ts)} runs, median {1e3 * float(np.median(ts)):.3f} ms", flush=True)
    sys.exit(0)

def timed(h, reps=7):
    ts = []; chi = None
    for _ in range(reps):
        h.restore_state(); t = time.perf_counter(); chi = h.optimize(10)["chi2"]; ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts)), chi

if "dummy_first" in sys.argv:
    h0 = HipSolver(fp, RK)          # a live handle before any other: no handle of this process ever instantiates a hipGraph
h1 = make()
solo, chi_solo = timed(h1)
print(f"{shape} {OPTS}: one handle {solo:.3f} ms per optimize(10)", flush=True)
h2 = make()
a, _ = timed(h1); b, chib = timed(h2)
print(f"  with a second live (idle) handle: first {a:.3f} ms ({a / solo:.2f} x), the second one itself {b:.3f} ms ({b / solo:.2f} x), bit-identical {np.array_equal(chib, chi_solo)}", flush=True)
h3 = make(); h4 = make()
a, _ = timed(h1); d, _ = timed(h4)
print(f"  with four live handles: first {a:.3f} ms ({a / solo:.2f} x), fourth {d:.3f} ms ({d / solo:.2f} x)", flush=True)
for group in ([h1, h2], [h1, h2, h3, h4]):
    res = [None] * len(group)
    bar = threading.Barrier(len(group) + 1)
    def work(i):
        h = group[i]; ts = []; chi = None
        bar.wait()
        for _ in range(7):
            h.restore_state(); t = time.perf_counter(); chi = h.optimize(10)["chi2"]; ts.append(time.perf_counter() - t)
        res[i] = (1e3 * float(np.median(ts)), chi, h.counter("pcg_iterations_plain_launches"), h.counter("pcg_iterations_enqueued"))
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(group))]
    for t in th: t.start()
    bar.wait(); t0 = time.perf_counter()
    for t in th: t.join()
    wall = time.perf_counter() - t0
    rate = len(group) * 7 * 10 * E / wall
    print(f"  {len(group)} graphs concurrently: per-graph median " + ", ".join(f"{r[0]:.3f}" for r in res) + f" ms; aggregate {rate / 1e6:.0f} M edge-iterations/s = {rate / (10 * E / (solo * 1e-3)):.2f} x one graph; "
          + f"bit-identical to solo: {all(np.array_equal(r[1], chi_solo) for r in res)}; plain-launch / enqueued iterations so far " + ", ".join(f"{r[2]}/{r[3]}" for r in res), flush=True)
h2.close(); h3.close(); h4.close()
a, _ = timed(h1)
print(f"  after closing the others: {a:.3f} ms ({a / solo:.2f} x)", flush=True)
