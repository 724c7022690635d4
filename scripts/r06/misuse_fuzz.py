"""Random call sequences on the C ABI (stage API, uploads, snapshots, options, queries -- in any order, also before a graph exists): every call must
either work or come back as a reported error; afterwards a fresh upload + optimize on the same handle still follows the oracle."""
import os, sys, copy, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
faulthandler.enable()
import numpy as np
from cuba_amd import capi
from cuba_amd.capi import CubaHipError
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_ba
from oracle.oracle import OracleSolver
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(9)
rk = ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815))))
graphs = [flatten(synth_ba(P, L, E, seed=P)) for P, L, E in ((10, 100, 300), (40, 600, 2400), (150, 3000, 12000))]
gold = [OracleSolver(fp, rk).optimize(4)["chi2"] for fp in graphs]
errors = 0; calls = 0; bad = 0
for rd in range(rounds):
    h = capi.HipSolver(None, rk, precision="f64" if rng.random() < 0.8 else "f32")
    f32 = h.scalar_size == 4
    have = None
    ops = ["set_graph", "set_graph2", "build_structure", "compute_errors", "build_system", "max_diagonal", "set_lambda", "restore_diagonal", "schur", "solve_reduced",
           "back_substitute", "solve", "update", "compute_scale", "push", "pop", "snapshot", "restore", "optimize", "optimize0", "state", "set_state", "chi_squares",
           "chi_two_step", "counters", "pcg_history", "hsc", "array", "hint", "option", "bad_option", "schur_parts", "assemble", "time_kernels", "profile"]
    for step in range(int(rng.integers(20, 60))):
        op = ops[int(rng.integers(len(ops)))]; calls += 1
        try:
            if op == "set_graph": have = int(rng.integers(3)); h.set_graph(graphs[have])
            elif op == "set_graph2": have = int(rng.integers(3)); h.set_graph(graphs[have], two_step=rng.choice([True, "begin_only"]))
            elif op == "build_structure": h.build_structure()
            elif op == "compute_errors": h.compute_errors()
            elif op == "build_system": h.build_system()
            elif op == "max_diagonal": h.max_diagonal()
            elif op == "set_lambda": h.set_lambda(float(rng.choice([1e-3, 1.0, 1e3, 0.0, -1.0, np.nan])))
            elif op == "restore_diagonal": h.restore_diagonal()
            elif op == "schur": h.schur()
            elif op == "solve_reduced": h.solve_reduced()
            elif op == "back_substitute": h.back_substitute()
            elif op == "solve": h.solve()
            elif op == "update": h.update()
            elif op == "compute_scale": h.compute_scale(1.0)
            elif op == "push": h.push()
            elif op == "pop": h.pop()
            elif op == "snapshot": h.snapshot_state(int(rng.integers(-1, 70)))
            elif op == "restore": h.restore_state(int(rng.integers(-1, 70)))
            elif op == "optimize": h.optimize(int(rng.integers(1, 4)))
            elif op == "optimize0": h.optimize(0)
            elif op == "state": h.state()
            elif op == "set_state":
                if have is not None: q, t, X = h.state(); h.set_state(q, t, X + rng.normal(0, 1e-3, X.shape))
            elif op == "chi_squares": h.chi_squares()
            elif op == "chi_two_step": h.chi_squares_two_step()
            elif op == "counters": h.counters(); h.counter("lm_trials")
            elif op == "pcg_history": h.pcg_history()
            elif op == "hsc": h.hsc()
            elif op == "array": h.array(str(rng.choice(["xp", "xl", "bp", "bsc", "lm_sys", "nonsense"])))
            elif op == "hint": h.hint_unchanged(bool(rng.integers(2)), bool(rng.integers(2)))
            elif op == "option": h.set_option(str(rng.choice(["pcg_tol", "pcg_max_iter", "reduced_solver", "pcg_aggregate", "coarse_linear", "precond_fp32", "mixed_precision", "heuristics", "direct_fallback", "device_setup", "pcg_graph"])),
                                               float(rng.choice([0, 1, 2, 8, 1e-9, 1e-3, -1, 1e9, np.nan])))
            elif op == "bad_option": h.set_option("no_such_option", 1.0)
            elif op == "schur_parts": h.schur_parts()
            elif op == "assemble": h.assemble()
            elif op == "time_kernels": h.time_kernels(2)
            elif op == "profile": h.profile()
        except CubaHipError:
            errors += 1
        except (ValueError, KeyError, AssertionError, AttributeError, TypeError):      # (the Python wrapper before a graph exists, its own argument checks)
            errors += 1
    # the handle must still be good for an ordinary job
    try:
        for k, v in (("pcg_tol", 1e-4 if f32 else 1e-7), ("pcg_max_iter", 0), ("reduced_solver", 0), ("pcg_aggregate", -1), ("coarse_linear", 1), ("precond_fp32", 1), ("mixed_precision", 0),
                     ("heuristics", 1), ("direct_fallback", 1), ("device_setup", 1), ("pcg_graph", 0)):
            try: h.set_option(k, v)
            except CubaHipError: pass
        i = int(rng.integers(3)); h.set_graph(graphs[i]); r = h.optimize(4)["chi2"]
        d = float(np.max(np.abs(r - gold[i][:len(r)]) / gold[i][:len(r)])) if len(r) == len(gold[i]) else 1.0
        if d > (1e-3 if f32 else 1e-6): bad += 1; print(f"FAIL round {rd}: after the random calls the handle's run differs from the oracle by {d:.2e} (lengths {len(r)} {len(gold[i])})", flush=True)
    except CubaHipError as e:
        bad += 1; print(f"FAIL round {rd}: the handle refuses an ordinary job afterwards: {e}", flush=True)
    h.close()
print(f"{rounds} handles, {calls} random calls ({errors} came back as reported errors), no crash; {bad} handles unusable or wrong afterwards", flush=True)
