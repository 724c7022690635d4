"""Pins against the REFERENCE ITSELF: the reference's own kernels (oracle/_ref, built from
/root/reference/src/cuda_block_solver.cu through a name shim, see oracle/ref_build/) run one LM trial on the
MI355X; both the CPU oracle and the HIP path must reproduce every stage output.  Skipped when oracle/_ref was
not built (it can only be built where the reference checkout exists)."""
import numpy as np
import pytest

from conftest import RK_HUBER, RK_NONE, RK_TUKEY, with_fixed
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_ba
from oracle import ref_kernels

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref_kernels.available(), reason="oracle/_ref not built")]

TOL = 1e-9


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def sym6(rows):
    a = np.asarray(rows); M = np.zeros((len(a), 3, 3))
    for k, (i, j) in enumerate([(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]):
        M[:, i, j] = a[:, k]; M[:, j, i] = a[:, k]
    return M


def cases():
    g = synth_ba(40, 600, 2400, seed=1)
    yield "huber", flatten(g), RK_HUBER
    yield "none", flatten(g), RK_NONE
    yield "tukey", flatten(g), RK_TUKEY
    yield "fixed", flatten(with_fixed(g, fixed_pose_rows=[3, 4, 20], fixed_lm_rows=list(range(0, 300, 7)))), RK_HUBER
    yield "mono_only", flatten(synth_ba(30, 400, 1600, seed=4, stereo_frac=0.0)), RK_HUBER
    yield "stereo_only", flatten(synth_ba(30, 400, 1600, seed=4, stereo_frac=1.0)), RK_HUBER


@pytest.mark.parametrize("name,fp,rk", list(cases()), ids=[c[0] for c in cases()])
def test_oracle_and_hip_match_reference_kernels(name, fp, rk):
    from cuba_amd.capi import HipSolver
    from oracle.oracle import OracleSolver
    o = OracleSolver(fp, rk)
    chi0 = o.compute_errors(); o.build_system()
    md = o.max_diagonal(); lam = 1e-5 * md
    o.set_lambda(lam); assert o.solve()
    rp, ci, vo = o.hsc()
    xp = o.array("xp")
    ref = ref_kernels.run_trial(fp, rk, rp, ci, lam, xp)

    # ---- CPU oracle vs reference kernels (this is what pins the oracle) --------------------------------
    assert ref["chi2"] == pytest.approx(chi0, rel=1e-12)
    assert ref["maxdiag"] == pytest.approx(md, rel=1e-12)
    Hpp = o.array("Hpp").reshape(-1, 6, 6).copy()
    for k in range(6):
        Hpp[:, k, k] -= lam                                       # oracle holds Hpp + lambda I after set_lambda
    Hll = o.array("Hll").reshape(-1, 3, 3).copy()
    for k in range(3):
        Hll[:, k, k] -= lam
    assert rel(ref["Hpp"].reshape(-1, 6, 6), Hpp) < TOL and rel(ref["bp"], o.array("bp")) < TOL
    assert rel(ref["Hll"].reshape(-1, 3, 3), Hll) < TOL and rel(ref["bl"], o.array("bl")) < TOL
    assert rel(ref["bsc"], o.array("bsc")) < TOL
    _, _, vraw = o.hsc_values_raw()
    assert rel(ref["hsc"], vraw) < TOL
    assert rel(ref["invHll"], o.array("invHll")) < 1e-8
    assert rel(ref["xl"], o.array("xl")) < 1e-8
    assert ref["scale"] == pytest.approx(o.compute_scale(lam), rel=1e-9)
    o.update()
    for a, b in zip((ref["q"], ref["t"], ref["Xw"]), o.state()):
        assert np.abs(a - b).max() < 1e-10
    assert ref["chi2_after"] == pytest.approx(o.compute_errors(), rel=1e-10)
    assert rel(ref["chi_per_edge"], o.chi_squares()) < 1e-9

    # ---- HIP path vs reference kernels ---------------------------------------------------------------------
    h = HipSolver(fp, rk, pcg_tol=1e-11)
    assert h.compute_errors() == pytest.approx(ref["chi2"], rel=1e-12)
    assert h.max_diagonal() == pytest.approx(ref["maxdiag"], rel=1e-12)
    lm = h.array("lm_sys").reshape(-1, 9)
    assert rel(sym6(lm[:, :6]), ref["Hll"].reshape(-1, 3, 3)) < TOL and rel(lm[:, 6:], ref["bl"].reshape(-1, 3)) < TOL
    assert rel(h.array("bp"), ref["bp"]) < TOL
    h.set_lambda(lam); h.schur()
    hrp, hci, hv = h.hsc()
    assert np.array_equal(hrp, rp) and np.array_equal(hci, ci)
    refv = ref["hsc"].reshape(-1, 6, 6).transpose(0, 2, 1)
    diag = np.zeros(len(ci), bool); diag[rp[:-1]] = True
    assert rel(hv[~diag], refv[~diag]) < TOL
    iu = np.triu_indices(6)
    assert rel(hv[diag][:, iu[0], iu[1]] + lam * (iu[0] == iu[1]), refv[diag][:, iu[0], iu[1]]) < TOL
    assert rel(h.array("bsc"), ref["bsc"]) < TOL
    assert rel(sym6(h.array("lm_sys").reshape(-1, 9)[:, :6]), ref["invHll"].reshape(-1, 3, 3)) < 1e-8
    assert h.solve_reduced()
    assert rel(h.array("xp"), xp) < 1e-6
    h.back_substitute()
    assert rel(h.array("xl"), ref["xl"]) < 1e-6
    assert h.compute_scale(lam) == pytest.approx(ref["scale"], rel=1e-7)
    h.update()
    for a, b in zip(h.state(), (ref["q"], ref["t"], ref["Xw"])):
        assert np.abs(a - b).max() < 1e-8
    assert h.compute_errors() == pytest.approx(ref["chi2_after"], rel=1e-8)
