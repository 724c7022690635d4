"""Symbolic phase of the exact reduced solve (csrc/ba_direct.hip: sparse_chol_plan -- minimum-degree ordering of 5-pose segments with
multiple elimination, fill, elimination-tree levels, gather lists; the seat of the reference's SparseLinearSolver::initialize,
src/cuda_linear_solver.cpp:278-348), checked on the CPU: a system is solved THROUGH the plan by a numpy restatement of the numeric
kernels (tests/sparse_chol_emulator.py) and compared with LAPACK.  No GPU involved: the C-ABI hook runs on the host."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
from sparse_chol_emulator import random_spd_blocks, solve_through_plan  # noqa: E402

from cuba_amd.capi import sparse_plan  # noqa: E402


def band_pattern(P, half, closures=()):
    rows = [set([i]) | set(range(i, min(P, i + half + 1))) for i in range(P)]
    for a, b, w in closures:                      # poses a .. a + w see poses b .. b + w again
        for i in range(a, min(P, a + w)):
            for j in range(b, min(P, b + w)):
                rows[min(i, j)].add(max(i, j))
    row_ptr = [0]; col_ind = []
    for i in range(P):
        col_ind += sorted(rows[i]); row_ptr.append(len(col_ind))
    return np.array(row_ptr, dtype=np.int32), np.array(col_ind, dtype=np.int32)


CASES = {
    "one_pose": lambda: band_pattern(1, 0),
    "one_tile": lambda: band_pattern(5, 4),
    "short_last_segment": lambda: band_pattern(13, 3),
    "dense_40": lambda: band_pattern(40, 40),
    "band_120": lambda: band_pattern(120, 9),
    "band_loop_closure": lambda: band_pattern(203, 7, closures=[(0, 150, 40)]),
    "block_diagonal": lambda: band_pattern(61, 0),
    "two_closures": lambda: band_pattern(300, 5, closures=[(10, 200, 12), (60, 280, 15)]),
}


@pytest.mark.parametrize("slack", [-1, 0, 4])
@pytest.mark.parametrize("name", sorted(CASES))
def test_plan_solves_the_system(name, slack):
    row_ptr, col_ind = CASES[name]()
    plan = sparse_plan(row_ptr, col_ind, slack=slack)
    P = len(row_ptr) - 1
    T = (P + 4) // 5
    assert plan["T"] == T and sorted(plan["posOfSeg"].tolist()) == list(range(T))
    assert plan["nblk"] == len(col_ind) and len(plan["lvlTiles"]) == plan["nTiles"] and len(plan["lvlCols"]) == T
    assert sorted(plan["lvlTiles"].tolist()) == list(range(plan["nTiles"])) and sorted(plan["lvlCols"].tolist()) == list(range(T))
    # every column: the diagonal tile first, then strictly ascending rows below it
    for k in range(T):
        r = plan["rowIdx"][plan["colPtr"][k]:plan["colPtr"][k + 1]]
        assert r[0] == k and np.all(np.diff(r) > 0)
    assert np.all(np.diff(plan["gPtr"]) % 4 == 0)              # the kernel takes four gather entries per trip
    rng = np.random.default_rng(len(col_ind))
    A = random_spd_blocks(row_ptr, col_ind, rng)
    b = rng.normal(size=6 * P)
    x = solve_through_plan(plan, row_ptr, col_ind, A, b)
    ref = np.linalg.solve(A, b)
    assert np.abs(x - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    # deterministic: a function of the pattern
    again = sparse_plan(row_ptr, col_ind, slack=slack)
    assert all(np.array_equal(plan[k], again[k]) for k in plan if isinstance(plan[k], np.ndarray))


def test_fill_and_depth_on_a_trajectory_with_a_loop_closure():
    """What the ordering is for: a trajectory whose last stretch sees its first one again.  The natural (band) order fills every row of
    the revisit over the whole lap; minimum degree keeps the factor at a few tiles per column, and multiple elimination makes the tree
    shallow (a launch per level instead of one per tile column)."""
    P = 1000
    row_ptr, col_ind = band_pattern(P, 18, closures=[(0, 770, 230)])
    T = P // 5
    nat_rows_revisit = (230 // 5) * (770 // 5)                       # tiles the band order would fill for the revisit alone
    p0 = sparse_plan(row_ptr, col_ind, slack=0)
    p4 = sparse_plan(row_ptr, col_ind, slack=4)
    pa = sparse_plan(row_ptr, col_ind, slack=-1)
    for p in (p0, p4, pa):
        assert p["nTiles"] < nat_rows_revisit and p["nTiles"] < 40 * T
    assert p4["nLevels"] < p0["nLevels"] and p4["nLevels"] <= T // 2
    assert pa["nLevels"] <= p0["nLevels"]
    print(f"\n[sparse plan] {P} poses, {T} tile columns: slack 0 -> {p0['nTiles']} tiles / {p0['nLevels']} levels, slack 4 -> {p4['nTiles']} / {p4['nLevels']}, "
          f"automatic (slack {pa['slack']}) -> {pa['nTiles']} / {pa['nLevels']}")


def test_dense_pattern_fills_every_tile():
    row_ptr, col_ind = band_pattern(60, 60)
    plan = sparse_plan(row_ptr, col_ind)
    assert plan["nTiles"] == 12 * 13 // 2                            # dense: every tile on or below the diagonal
