"""A/B of solver options with ONE live handle at a time (several live handles share the runtime's few hardware queues and slow each
other down: the second handle of a process runs a KITTI-07-sized graph at half speed), the variant order repeated so that drift shows.
   python scripts/option_ab_seq.py kitti00 "" fused_tail=0 precond_fp32=0 ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cuba_amd.capi import HipSolver
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_named

RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
shape = sys.argv[1]
fp = flatten(synth_named(shape))
variants = [dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in a.split(",") if kv) for a in sys.argv[2:]] or [{}]
res = {i: [] for i in range(len(variants))}
info = {}
for rnd in range(2):
    for i, v in enumerate(variants):
        h = HipSolver(fp, RK, **v)
        h.optimize(1)
        q1, t1, X1 = h.state()
        h.optimize(10)
        for rep in range(6):
            h.set_state(q1, t1, X1); c0 = h.counters()
            t = time.perf_counter(); got = h.optimize(10)["chi2"]; res[i].append(time.perf_counter() - t)
        c = h.counters()
        info[i] = (c["pcg_iterations"] - c0["pcg_iterations"], c["pcg_iterations_enqueued"] - c0["pcg_iterations_enqueued"], got[-1])
        h.close()
for i, v in enumerate(variants):
    ts = np.array(res[i]) * 1e3
    print("%-8s %-44s min %.3f  median %.3f ms   round medians %.3f / %.3f   pcg %d enq %d chi2 %.6f" % (
        shape, v, ts.min(), np.median(ts), np.median(ts[:6]), np.median(ts[6:]), *info[i]), flush=True)
