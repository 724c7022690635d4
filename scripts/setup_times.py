"""Set-up cost per shape: cuba_hip_set_graph + cuba_hip_build_structure for a NEW topology (second call on a warm handle,
structure cache off), device pipeline (default) against the host pipeline (device_setup = 0), and the 10-iteration LM wall.
   python scripts/setup_times.py kitti00 s2m g4m"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CUBA_HIP_NO_STRUCTURE_CACHE"] = "1"
import numpy as np
from cuba_amd.synth import synth_named
from cuba_amd.graph import flatten
from cuba_amd.capi import HipSolver
RK = ((1, np.sqrt(5.991)), (1, np.sqrt(7.815)))
for name in sys.argv[1:] or ["kitti00"]:
    fp = flatten(synth_named(name))
    line = f"{name:8s} P {fp.Pt} L {fp.Lt} E {fp.E}:"
    for label, opts in (("device", {}), ("host", dict(device_setup=0))):
        h = HipSolver(fp, RK, **opts); h.build_structure(); h.optimize(2)
        ts = []
        for _ in range(5):
            t = time.perf_counter(); h.set_graph(fp); h.build_structure(); ts.append(time.perf_counter() - t)
        t = time.perf_counter(); n = len(h.optimize(10)["chi2"]); dt = time.perf_counter() - t
        line += f"  {label} set-up {min(ts)*1e3:.2f} ms (median {np.median(ts)*1e3:.2f})"
        if label == "device":
            line += f", 10 LM iterations {dt*1e3:.1f} ms ({n} done, {int(np.abs(h.pcg_history()[0][-10:]).sum())} PCG iterations);"
        h.close()
    print(line, flush=True)
