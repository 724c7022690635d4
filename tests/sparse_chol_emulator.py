"""numpy restatement of the NUMERIC phase of the exact reduced solve (csrc/ba_direct.hip: schol_fill_*, schol_factor_level_kernel,
schol_back_level_kernel, schol_extract_kernel), driven by the arrays of the library's symbolic phase (capi.sparse_plan).  Test
infrastructure: it lets the CPU suite check the plan -- ordering, fill, gather lists, levels, block -> tile map -- without a GPU, by
solving a system through it and comparing with LAPACK.  Every step asserts the dependency the level schedule promises (a gathered tile
was finished at an earlier level)."""
import numpy as np

TP, TS = 5, 32


def random_spd_blocks(row_ptr, col_ind, rng, cond_shift=1.0):
    """dense symmetric positive-definite matrix with exactly the given upper-triangular 6 x 6 block pattern"""
    P = len(row_ptr) - 1
    A = np.zeros((6 * P, 6 * P))
    for i in range(P):
        for k in range(row_ptr[i], row_ptr[i + 1]):
            j = col_ind[k]
            B = rng.normal(size=(6, 6))
            if i == j:
                B = B + B.T
            A[6 * i:6 * i + 6, 6 * j:6 * j + 6] = B
            A[6 * j:6 * j + 6, 6 * i:6 * i + 6] = B.T
    A += np.diag(np.abs(A).sum(1) + cond_shift)
    return A


def solve_through_plan(plan, row_ptr, col_ind, A, b):
    P = len(row_ptr) - 1
    T, nT = plan["T"], plan["nTiles"]
    pos, colPtr, rowIdx = plan["posOfSeg"], plan["colPtr"], plan["rowIdx"]
    tiles = np.zeros((nT + 1, TS, TS))          # [tile][row][col]
    # fill (schol_fill_blocks_kernel / schol_fill_rhs_kernel)
    for bi in range(P):
        for k in range(row_ptr[bi], row_ptr[bi + 1]):
            bj = col_ind[k]
            tt = int(plan["blkTile"][k]); tile, tr = tt & 0x3fffffff, tt >> 30
            blk = A[6 * bi:6 * bi + 6, 6 * bj:6 * bj + 6]           # element (r, c)
            for r in range(6):
                for c in range(6):
                    if bi == bj and c < r:
                        continue
                    li, lj = 6 * (bi % TP) + r, 6 * (bj % TP) + c
                    row, col = (li, lj) if tr else (lj, li)
                    tiles[tile, row, col] = blk[r, c]
    y = np.zeros(TS * T)
    for seg in range(T):
        k = pos[seg]
        for i in range(TS):
            p = TP * seg + i // 6
            if i < 6 * TP and p < P:
                y[TS * k + i] = b[6 * p + i % 6]
            else:
                tiles[colPtr[k], i, i] = 1.0
    colOf = np.zeros(nT, dtype=int)
    for k in range(T):
        colOf[colPtr[k]:colPtr[k + 1]] = k
    done = np.zeros(nT + 1, dtype=bool); done[nT] = True          # the zero tile
    G = plan["gather"].reshape(-1, 4)
    Ldiag = {}
    for l in range(plan["nLevels"]):
        lo, hi = plan["lvlPtr"][l], plan["lvlPtr"][l + 1]
        new = {}
        for t in plan["lvlTiles"][lo:hi]:
            j = colOf[t]; t0 = colPtr[j]
            D = tiles[t0].copy(); F = tiles[t].copy(); v = np.zeros(TS)
            D = np.tril(D) + np.tril(D, -1).T
            for ta, tb, k, _ in G[plan["gPtr"][t]:plan["gPtr"][t + 1]]:
                assert done[ta] and done[tb], "gathered tile not finished at an earlier level"
                D -= tiles[tb] @ tiles[tb].T
                if t != t0:
                    F -= tiles[ta] @ tiles[tb].T
                else:
                    v += tiles[tb] @ y[TS * k:TS * k + TS]
            Lj = np.linalg.cholesky(D)
            if t == t0:
                new[t] = ("diag", Lj, np.linalg.solve(Lj, y[TS * j:TS * j + TS] - v), j)
            else:
                new[t] = ("off", np.linalg.solve(Lj, F.T).T)
        for t, rec in new.items():          # (a level's results become visible together, like a kernel launch's)
            if rec[0] == "diag":
                Ldiag[rec[3]] = rec[1]; y[TS * rec[3]:TS * rec[3] + TS] = rec[2]
            else:
                tiles[t] = rec[1]
            done[t] = True
    assert done.all()
    xdone = np.zeros(T, dtype=bool)
    for l in range(plan["nLevels"] - 1, -1, -1):
        cols = plan["lvlCols"][plan["lvlColPtr"][l]:plan["lvlColPtr"][l + 1]]
        new = {}
        for j in cols:
            v = np.zeros(TS)
            for t in range(colPtr[j] + 1, colPtr[j + 1]):
                i = rowIdx[t]
                assert xdone[i], "backward substitution reads a column of a later level"
                v += tiles[t].T @ y[TS * i:TS * i + TS]
            new[j] = np.linalg.solve(Ldiag[j].T, y[TS * j:TS * j + TS] - v)
        for j, xj in new.items():
            y[TS * j:TS * j + TS] = xj; xdone[j] = True
    x = np.zeros(6 * P)
    for p in range(P):
        x[6 * p:6 * p + 6] = y[TS * pos[p // TP] + 6 * (p % TP):TS * pos[p // TP] + 6 * (p % TP) + 6]
    return x
